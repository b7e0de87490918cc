"""launches around the large idle gaps of a rocprofv3 kernel trace:  python scripts/trace_gaps.py <kernel_trace.csv> [min gap us = 300] [context = 3]
(what runs before and after every gap longer than the threshold: finds host-side work between launches inside a solve)"""
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))), key=lambda t: t[0])
thr = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 300e3
ctx = int(sys.argv[3]) if len(sys.argv) > 3 else 3
t0 = rows[0][0]
n = 0
for i in range(1, len(rows)):
    gap = rows[i][0] - rows[i - 1][1]
    if gap > thr:
        n += 1
        print(f"--- gap {gap/1e3:9.1f} us at t = {(rows[i][0]-t0)/1e6:9.3f} ms")
        for k in range(max(0, i - ctx), min(len(rows), i + ctx)):
            print(f"   {'>' if k == i else ' '} {(rows[k][0]-t0)/1e6:9.3f} ms  {(rows[k][1]-rows[k][0])/1e3:8.1f} us  {rows[k][2][:110]}")
print(f"{n} gaps longer than {thr/1e3:.0f} us; span {(rows[-1][1]-t0)/1e6:.1f} ms")

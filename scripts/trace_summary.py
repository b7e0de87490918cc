"""summary of a rocprofv3 --kernel-trace CSV: per kernel (template arguments kept) calls / total / average, and how busy the GPU was between the
first and the last launch of the window (sum of kernel durations / wall span) -- what is NOT kernel time is launch gaps and host round trips.
   python scripts/trace_summary.py <kernel_trace.csv> [last_fraction]      last_fraction: analyse only the last part of the trace (default 0.5:
   skips generation / warm-up launches)"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * (1 - frac)):]
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
per = collections.defaultdict(lambda: [0, 0])
busy = 0
for r in rows:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    k = r["Kernel_Name"]
    k = k[:150]
    per[k][0] += 1; per[k][1] += d
    busy += d
print(f"window: {len(rows)} launches, span {(t1 - t0) / 1e6:.3f} ms, kernel time {busy / 1e6:.3f} ms = {busy / (t1 - t0):.3f} of the span")
for k, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{t / 1e3:10.1f} us total {c:7d} calls {t / c / 1e3:8.2f} us avg  {100.0 * t / (t1 - t0):5.1f} % of span  {k}")

# idle time in front of each kernel name: the gap between the end of the previous launch (by start order) and this launch's start --
# kernel boundaries, no-op launches' dispatch and host round trips all show here
gaps = collections.defaultdict(lambda: [0, 0])
prev_end = None
for r in rows:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if prev_end is not None and st > prev_end:
        k = r["Kernel_Name"][:110]
        gaps[k][0] += 1; gaps[k][1] += st - prev_end
    prev_end = en if prev_end is None else max(prev_end, en)
tot = sum(v[1] for v in gaps.values())
print(f"\nidle {tot / 1e6:.3f} ms = {tot / (t1 - t0):.3f} of the span, by the kernel that follows the gap:")
for k, (c, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"{t / 1e3:10.1f} us total {c:7d} gaps {t / c / 1e3:8.2f} us avg  {k}")

# optional: a window of consecutive launches (python trace_summary.py trace.csv 0.7 --window N): offset from the first one, duration, gap to the previous end
if "--window" in sys.argv:
    n = int(sys.argv[sys.argv.index("--window") + 1])
    mid = len(rows) // 2
    w = rows[mid:mid + n]
    base = int(w[0]["Start_Timestamp"]); pe = None
    print(f"\n{n} consecutive launches from the middle of the window: start (us), duration (us), gap to the end of the previous launch (us; negative = overlap)")
    for r in w:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{(st - base) / 1e3:10.2f} {(en - st) / 1e3:8.2f} {((st - pe) / 1e3 if pe is not None else 0.0):8.2f}  {r['Kernel_Name'][:70]}")
        pe = en

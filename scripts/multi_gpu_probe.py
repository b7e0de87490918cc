"""First contact with a REAL multi-GPU node, rung by rung (nothing here has ever run on more than one physical GPU: DESIGN section 4):
   python scripts/multi_gpu_probe.py --gpus N [--rungs peer rccl procs] [--n 400]
 peer   one process, N devices (xm_problem_t.n_gpus = N, gpu_map = 0): direct peer writes -- peer access, fine-grained arenas, the transport's
        self-test (known contents through both read paths) -- then a small staircase solve against the single-GPU one
 rccl   the same context with the RCCL rung forced (xm_tuning_t.exchange = 3: ncclCommInitRank from N host threads, one rank per device)
 procs  N PROCESSES, one per device (what torch.distributed.run launches): xm_comm_init with an RCCL unique id, the IPC rendezvous of the peer
        transport on top of it, the same solve
Every rung prints one JSON line: what transport joined the ranks, why a faster one was given up, the optimum against the single-GPU solve.
A rung that fails prints the error and the next one still runs; exit code = number of failed rungs."""
import argparse, json, os, subprocess, sys, textwrap, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "xm-code_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--gpus", type=int, default=2)
    ap.add_argument("--rungs", nargs="*", default=["peer", "rccl", "procs"], choices=["peer", "rccl", "procs"])
    ap.add_argument("--n", type=int, default=400, help="cameras of the probe problem (dense view-graph Q that needs a rank escalation)")
    ap.add_argument("--dry-run", action="store_true", help="print what would run and exit (no GPU needed)")
    a = ap.parse_args()
    if a.dry_run:
        print(json.dumps(dict(gpus=a.gpus, rungs=a.rungs, n=a.n)))
        return 0
    import numpy as np, xmamd, xm_testlib as tl
    xmamd.require_gpu()
    ndev = xmamd.device_count()
    print(json.dumps(dict(devices_visible=ndev, gpus_requested=a.gpus)), flush=True)
    P = tl.gen_vg(a.n, deg=6, sigma=0.8, seed=a.n)
    Q = P["Q"]
    R1, s1, i1 = xmamd.solve_dense(Q, 6, 1e-9, 6.0)
    failed = 0
    for rung in a.rungs:
        t0 = time.time()
        try:
            if rung in ("peer", "rccl"):
                if ndev < a.gpus:
                    raise RuntimeError(f"{ndev} devices visible, {a.gpus} asked for (gpu_map = 0 needs one device per rank)")
                ctx = xmamd.Context(Q=Q, n_gpus=a.gpus, gpu_map=0, tuning=dict(exchange=2 if rung == "peer" else 3))
                kind, name, note = ctx.transport()
                R, s, info = ctx.solve(6, 1e-9, 6.0)
                ctx.close()
                out = dict(rung=rung, ok=True, transport=name, fallback=note or None, exchange=info.get("exchange"), rank=info["rank"], status=info["status"],
                           primal_rel=abs(info["primal"] - i1["primal"]) / abs(i1["primal"]), rotations=tl.rotation_parity(R, s, R1, s1), tcg_iters=info["tcg_iters"])
                out["ok"] = bool(info["status"] == i1["status"] and info["rank"] == i1["rank"] and out["primal_rel"] < 1e-8 and out["rotations"] < 1e-6)
            else:
                uid = (xmamd.C.c_char * 128)()
                xmamd._chk(xmamd.lib().xm_comm_unique_id(uid))
                idf = f"/tmp/xm_probe_uid_{os.getpid()}.bin"
                open(idf, "wb").write(uid.raw)
                code = textwrap.dedent(f"""
                    import sys, os, json
                    sys.path.insert(0, {os.path.join(ROOT, 'xm-code_amd')!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})
                    import numpy as np, xmamd, xm_testlib as tl
                    rank = int(sys.argv[1]); world = int(sys.argv[2]); uid = open(sys.argv[3], "rb").read()
                    rc = xmamd.lib().xm_comm_init(rank, world, rank, uid, None)
                    if rc != 0:
                        print(json.dumps(dict(ok=False, err=xmamd.lib().xm_last_error().decode()))); sys.exit(0)
                    P = tl.gen_vg({a.n}, deg=6, sigma=0.8, seed={a.n})
                    ctx = xmamd.Context(Q=P["Q"]); tr = ctx.transport(); R, s, info = ctx.solve(6, 1e-9, 6.0); ctx.close()
                    xmamd.lib().xm_comm_finalize()
                    print(json.dumps(dict(ok=True, primal=info["primal"], rank=info["rank"], status=info["status"], transport=tr[1], fallback=tr[2], exchange=info.get("exchange"))))
                """)
                env = dict(os.environ, NCCL_SOCKET_IFNAME="lo", HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN", XM_WATCHDOG_S="120")
                procs = [subprocess.Popen([sys.executable, "-c", code, str(r), str(a.gpus), idf], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
                         for r in range(a.gpus)]
                res = []
                for p in procs:
                    try:
                        o = p.communicate(timeout=600)[0].decode()
                    except subprocess.TimeoutExpired:
                        p.kill(); o = p.communicate()[0].decode() + "\nTIMEOUT"
                    js = [l for l in o.splitlines() if l.startswith("{")]
                    res.append(json.loads(js[-1]) if js else dict(ok=False, err=o[-800:]))
                os.remove(idf)
                ok = all(r.get("ok") for r in res) and all(abs(r["primal"] - i1["primal"]) <= 1e-8 * abs(i1["primal"]) and r["rank"] == i1["rank"] for r in res)
                out = dict(rung=rung, ok=bool(ok), ranks=res)
        except Exception as e:   # the ladder's own message (XM_ERR_COMM names both reasons) or a HIP error
            out = dict(rung=rung, ok=False, error=str(e)[-1200:])
        out["seconds"] = time.time() - t0
        failed += 0 if out["ok"] else 1
        print(json.dumps(out), flush=True)
    return failed


if __name__ == "__main__":
    sys.exit(main())

"""bench.py's promise for runs on several ranks: ONE JSON line in every case (CPU tests of the pieces that do not need a GPU)."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
        "config", "transport", "fallback", "error"}


def _run(code, env=None, timeout=60):
    e = dict(os.environ); e.update(env or {})
    return subprocess.run([sys.executable, "-c", textwrap.dedent(code)], cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)


def test_error_line_is_one_parseable_json_line_with_the_contract_keys():
    r = _run("""
        import argparse, bench
        bench._error_line(8, argparse.Namespace(steps=6, warmup=1), "some workload", "XM_ERR_COMM: peer writes failed; RCCL failed")
    """)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1
    d = json.loads(lines[0])
    assert set(d) == KEYS and d["value"] is None and d["n_gpus"] == 8 and "RCCL" in d["error"] and d["config"] == {"workload": "some workload"}


def test_watchdog_prints_the_error_line_and_ends_the_process():
    """a collective that never returns: after XM_BENCH_TIMEOUT_S rank 0 prints the error line (with the phase) and the process exits 3"""
    r = _run("""
        import argparse, time, bench
        bench._PHASE[0] = "timed solves"
        bench._arm_guards(0, 2, argparse.Namespace(steps=3, warmup=1), "w")
        time.sleep(30)
    """, env={"XM_BENCH_TIMEOUT_S": "0.5"})
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 3 and len(lines) == 1
    d = json.loads(lines[0])
    assert "watchdog" in d["error"] and "timed solves" in d["error"] and d["n_gpus"] == 2
    quiet = _run("""
        import argparse, time, bench
        bench._arm_guards(1, 2, argparse.Namespace(steps=3, warmup=1), "w")
        time.sleep(30)
    """, env={"XM_BENCH_TIMEOUT_S": "0.5"})
    assert quiet.returncode == 3 and not [l for l in quiet.stdout.splitlines() if l.startswith("{")]   # only rank 0 speaks


def test_rank0_reports_when_the_launcher_tears_the_job_down():
    r = _run("""
        import argparse, os, signal, time, bench
        bench._PHASE[0] = "communicator"
        bench._arm_guards(0, 8, argparse.Namespace(steps=3, warmup=1), "w")
        os.kill(os.getpid(), signal.SIGTERM)
        time.sleep(30)
    """)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 4 and len(lines) == 1 and "another rank failed" in json.loads(lines[0])["error"]


def test_first_multi_gpu_run_script_commands_parse():
    """scripts/first_multi_gpu_run.sh is what runs in the first minutes on a real N-GPU node (nothing in this tree has seen more than one
    physical GPU): its command list must at least parse here -- every script it names exists and accepts its flags at --help level."""
    import shlex, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run(["bash", os.path.join(root, "scripts", "first_multi_gpu_run.sh"), "--dry-run", "8", "multi_test"], capture_output=True, text=True, cwd=root)
    assert out.returncode == 0, out.stderr
    cmds = [l for l in out.stdout.splitlines() if l.startswith("python ")]
    assert len(cmds) >= 10 and any("torch.distributed.run" in c for c in cmds) and any("--exchange rccl" in c for c in cmds)
    seen = set()
    for c in cmds:
        parts = shlex.split(c)
        script = next((p for p in parts if p.endswith(".py")), None)
        if script is None:                      # python -m pytest ...
            continue
        assert os.path.exists(os.path.join(root, script)), script
        if script in seen or script.endswith("collect_profiles.py") or script.endswith("kbench_multi.py") or script.endswith("kbench_symw.py"):
            continue                            # (those three import the GPU library at module level or take no flags)
        seen.add(script)
        flags = [p for p in parts[parts.index(script) + 1:] if p.startswith("--")]
        h = subprocess.run([sys.executable, os.path.join(root, script), "--help"], capture_output=True, text=True, cwd=root)
        assert h.returncode == 0, (script, h.stderr[-400:])
        for fl in flags:
            assert fl in h.stdout, (script, fl)
    d = subprocess.run([sys.executable, os.path.join(root, "scripts", "multi_gpu_probe.py"), "--dry-run", "--gpus", "8"], capture_output=True, text=True, cwd=root)
    assert d.returncode == 0 and '"gpus": 8' in d.stdout

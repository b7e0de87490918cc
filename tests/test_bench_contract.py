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

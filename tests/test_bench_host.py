"""Host-side pieces of bench.py that need no GPU: the per-rank CPU affinity the launcher sets before any GPU call."""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ranks_get_disjoint_cpu_slices_before_any_gpu_call():
    code = f"""
import os, sys, json
sys.path.insert(0, {REPO!r})
import bench
before = sorted(os.sched_getaffinity(0))
mine = bench.pin_rank_to_cpus(int(sys.argv[1]), int(sys.argv[2]))
import torch
print(json.dumps({{"before": before, "mine": mine, "now": sorted(os.sched_getaffinity(0)), "gpu_initialised": bool(torch.cuda.is_initialized())}}))
"""
    import json
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 2:
        return
    got = []
    for r in range(2):
        out = subprocess.run([sys.executable, "-c", code, str(r), "2"], capture_output=True, text=True, check=True)
        got.append(json.loads(out.stdout.strip().splitlines()[-1]))
    a, b = got
    assert a["mine"] == a["now"] and b["mine"] == b["now"] and not a["gpu_initialised"]          # set before anything initialises the GPU
    assert set(a["mine"]) and set(b["mine"]) and not (set(a["mine"]) & set(b["mine"]))
    assert set(a["mine"]) | set(b["mine"]) <= set(allowed)
    one = subprocess.run([sys.executable, "-c", code, "0", "1"], capture_output=True, text=True, check=True)
    assert json.loads(one.stdout.strip().splitlines()[-1])["mine"] is None               # a single rank keeps every CPU
    env = dict(os.environ, LPI_NO_AFFINITY="1")
    off = subprocess.run([sys.executable, "-c", code, "0", "2"], capture_output=True, text=True, check=True, env=env)
    assert json.loads(off.stdout.strip().splitlines()[-1])["mine"] is None

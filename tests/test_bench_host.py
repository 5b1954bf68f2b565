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
mine = bench.pin_rank_to_cpus(int(sys.argv[1]), int(sys.argv[2]), enabled=len(sys.argv) < 4)
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
    off = subprocess.run([sys.executable, "-c", code, "0", "2", "off"], capture_output=True, text=True, check=True)      # bench.py --no-affinity
    assert json.loads(off.stdout.strip().splitlines()[-1])["mine"] is None


def _fake_sysfs(root, gpus, cpu_nodes=2):
    """A sysfs tree like an 8-GPU node's: KFD topology nodes (CPU nodes first, simd_count 0), DRM render nodes whose PCI device reports a NUMA node."""
    import os
    n = 0
    for c in range(cpu_nodes):
        d = root / "class/kfd/kfd/topology/nodes" / str(n)
        d.mkdir(parents=True)
        (d / "properties").write_text("cpu_cores_count 64\nsimd_count 0\ndrm_render_minor 0\n")
        n += 1
    for i, (minor, numa) in enumerate(gpus):
        d = root / "class/kfd/kfd/topology/nodes" / str(n)
        d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count 0\nsimd_count 1024\ndrm_render_minor {minor}\nlocation_id {256 * i}\n")
        dev = root / f"devices/pci{i}"
        dev.mkdir(parents=True)
        (dev / "numa_node").write_text(f"{numa}\n")
        (dev / "vendor").write_text("0x1002\n")
        rd = root / "class/drm" / f"renderD{minor}"
        rd.mkdir(parents=True)
        os.symlink(dev, rd / "device")
        n += 1


def test_gpu_numa_nodes_from_a_sysfs_tree(tmp_path, monkeypatch):
    """the GPU -> NUMA node map comes from sysfs (KFD topology order = HIP's device order), not from a guess about device order.  A
    platform whose GPUs alternate between the sockets — where the old 'first half on socket 0' rule is wrong — and the visible-devices filters."""
    sys.path.insert(0, REPO)
    import bench
    monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES", raising=False)
    _fake_sysfs(tmp_path, [(128 + i, i % 2) for i in range(8)])
    assert bench.gpu_numa_nodes(str(tmp_path)) == [0, 1, 0, 1, 0, 1, 0, 1]
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1,2,5")
    assert bench.gpu_numa_nodes(str(tmp_path)) == [1, 0, 1]
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    empty = tmp_path / "nothing"
    empty.mkdir()
    assert bench.gpu_numa_nodes(str(empty)) is None
    # a platform that does not say (-1) is reported as such: pin_rank_to_cpus then falls back to the even split
    other = tmp_path / "unknown"
    other.mkdir()
    _fake_sysfs(other, [(128, -1), (129, -1)])
    assert bench.gpu_numa_nodes(str(other)) == [-1, -1]

"""Register-spill guard for the kernels whose LDS-DMA pipelines a spill breaks (CPU test on the built objects, no GPU).

A scratch reload is a vector-memory operation followed by a compiler-inserted `s_waitcnt vmcnt(0)`: in the persistent GEMM and in the
streamed attention backward that wait drains the in-flight LDS-DMA of the next tile / of the slice ring.  The allocation is
fragile — removing one unused word from the kernel-argument struct once made the QuickGELU instantiations of the GEMM spill 56 bytes per
lane — so the build's own metadata is checked: `.private_segment_fixed_size` of every such kernel must be 0."""
import os
import re
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def _kernel_scratch(obj_name, tmp_path):
    src = os.path.join(REPO, "lpi_amd", "csrc", "build", obj_name)
    if not os.path.exists(src):
        pytest.skip(f"{obj_name} not built (run __graft_entry__.build())")
    if not os.path.exists(os.path.join(LLVM, "llvm-objdump")):
        pytest.skip("llvm-objdump not available")
    obj = shutil.copy(src, tmp_path / obj_name)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", str(obj)], check=True, capture_output=True)
    dev = [p for p in os.listdir(tmp_path) if "amdgcn" in p]
    assert dev, "no device code object in " + obj_name
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", str(tmp_path / dev[0])], check=True, capture_output=True, text=True).stdout
    out = {}
    for blk in notes.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        size = re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk)
        if name and size:
            out[name.group(1)] = int(size.group(1))
    assert out, "no kernel metadata parsed"
    return out


def test_persistent_gemm_instantiations_do_not_spill(tmp_path):
    ks = {k: v for k, v in _kernel_scratch("gemm256p.o", tmp_path).items() if "gemm256p_kernel" in k}
    assert len(ks) >= 20
    assert {k: v for k, v in ks.items() if v} == {}


def test_streamed_attention_backward_of_the_vision_tower_does_not_spill(tmp_path):
    ks = _kernel_scratch("attention4.o", tmp_path)
    # attn_bwd4_kernel<SV16, NKB = 7>: the vision tower's backward (Lp = 224) in bf16 and f16 mode — a scratch reload's vmcnt(0) would drain
    # the ring's LDS-DMA every iteration
    # ... <SV16, NKB, WIN>: the key-window launches of a sequence longer than 224 tokens (ViT-L/14: L = 273 = 224 keys on the NKB = 7 configuration +
    # 49 on the generic one, round 4) must not spill either
    vision = {k: v for k, v in ks.items() if re.search(r"attn_bwd4_kernelILb[01]ELi7ELi[01]E", k)}
    assert len(vision) == 4, sorted(ks)
    assert all(v == 0 for v in vision.values()), vision
    window_rest = {k: v for k, v in ks.items() if re.search(r"attn_bwd4_kernelILb[01]ELi0ELi2E", k)}
    assert len(window_rest) == 2 and all(v == 0 for v in window_rest.values()), window_rest

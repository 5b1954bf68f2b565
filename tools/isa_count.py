#!/usr/bin/env python3
"""Instruction mix of the kernels in one built object (CPU tool: llvm-objdump on the gfx950 code object inside lpi_amd/csrc/build/<name>.o).

usage: python3 tools/isa_count.py gemm256p.o [kernel-name regex]
prints per matching kernel: total instructions, v_pk_*, transcendental (v_exp / v_rcp / v_rsq / v_log / v_sqrt), v_cvt*, other VALU, MFMA,
ds_*, global/buffer, s_waitcnt / s_barrier — the numbers the epilogue arithmetic budget in DESIGN.md section 4 quotes.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    obj_name = sys.argv[1]
    pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
    tmp = tempfile.mkdtemp()
    obj = shutil.copy(os.path.join(REPO, "lpi_amd", "csrc", "build", obj_name), tmp)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", obj], check=True, capture_output=True, cwd=tmp)
    dev = [p for p in os.listdir(tmp) if "amdgcn" in p][0]
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", os.path.join(tmp, dev)], check=True, capture_output=True, text=True).stdout
    cur, counts = None, {}
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            cur = m.group(1)
            continue
        if cur is None or not line.startswith("\t") and not line.startswith(" "):
            continue
        ins = line.split()
        if not ins:
            continue
        op = ins[0]
        c = counts.setdefault(cur, {})
        if op.startswith("v_pk_"):
            k = "v_pk"
        elif re.match(r"v_(exp|rcp|rsq|log|sqrt|sin|cos)_", op):
            k = "trans"
        elif op.startswith("v_cvt"):
            k = "v_cvt"
        elif op.startswith("v_mfma") or op.startswith("v_smfma"):
            k = "mfma"
        elif op.startswith("v_accvgpr"):
            k = "accvgpr"
        elif op.startswith("v_"):
            k = "valu"
        elif op.startswith("ds_"):
            k = "ds"
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            k = "vmem"
        elif op in ("s_waitcnt", "s_barrier"):
            k = op
        else:
            k = "salu/other"
        c[k] = c.get(k, 0) + 1
    shutil.rmtree(tmp)
    keys = ["v_pk", "trans", "v_cvt", "valu", "accvgpr", "mfma", "ds", "vmem", "s_waitcnt", "s_barrier", "salu/other"]
    for name, c in counts.items():
        if not pat.search(name):
            continue
        print(f"{sum(c.values()):7d} total | " + " ".join(f"{k} {c.get(k, 0)}" for k in keys) + f" | {name[:90]}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (one directory per pass, any number of counters per pass) into profiles/<tag>_pmc.json.

Per kernel (launch-averaged):
  hbm_mb_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes: FETCH_SIZE/WRITE_SIZE are in KiB and, on gfx950, FETCH_SIZE reports
                      exactly half of the bytes of a wide coalesced streaming read (MI355X_MICROARCH.md, HBM section); the two counters
                      are collected in separate passes (they do not fit one pass); Infinity-Cache hits are included in both;
  mfma_busy_frac    = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024): the guide's MfmaUtil expression with rocprofv3's
                      GRBM_GUI_ACTIVE (a sum over the 8 XCDs) brought back to one XCD's busy cycles, 1024 = SIMDs on the chip;
  clock_ghz         = GRBM_GUI_ACTIVE / 8 / kernel duration (guide, 'DVFS give-back'; reads high on dispatches < 0.3 ms);
  the SQ_WAIT_* / SQ_ACTIVE_INST_ANY / SQ_LDS_* sums as collected.

usage: tools/pmc_summary.py <out.json> <dtype> <tuning as comma list> <lib_version> <command string> <pass_dir> [<pass_dir> ...]
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(n):
    m = re.match(r"_ZN\d+_GLOBAL__N_1(\d+)", n)      # rocprofv3 leaves some instantiations (fp16 template arguments) mangled
    if m:
        k = int(m.group(1))
        n = n[m.end():m.end() + k]
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"<.*", "", n).replace("void ", "")


def load_pass(d):
    """-> {kernel: {counter: [launches, sum]}}, {kernel: [launches, total duration ns]} for one pass directory."""
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    dur = collections.defaultdict(lambda: [0, 0.0])
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            a = agg[short(r["Kernel_Name"])][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            t = dur[short(r["Kernel_Name"])]
            t[0] += 1
            t[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return agg, dur


def main():
    out_path, dtype, tuning, libv, cmd = sys.argv[1:6]
    counters = collections.defaultdict(dict)      # kernel -> counter -> (launches, sum)
    durs = {}
    for d in sys.argv[6:]:
        agg, dur = load_pass(d)
        for k, cs in agg.items():
            for c, (n, v) in cs.items():
                counters[k][c] = (n, v)
                if c == "GRBM_GUI_ACTIVE" and k in dur:
                    durs[k] = dur[k]
    # the workload the passes ran (bench.py attaches the summary only to a record of the SAME model / batch / depth): read off the command line
    def opt(name, default, conv=str):
        m = re.search(r"--" + name + r"[ =](\S+)", cmd)
        return conv(m.group(1)) if m else default
    workload = {"model": opt("model", "ViT-B/16"), "batch": opt("batch", 256, int), "depth": opt("depth", 3, int)}
    out = {"command": cmd, "workload": workload, "dtype": dtype, "tuning": [int(x) for x in tuning.split(",") if x], "lib_version": int(libv),
           "formulas": {"hbm_mb_per_launch": "(2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes (gfx950 FETCH_SIZE half-count corrected; separate passes)",
                        "mfma_busy_frac": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)",
                        "clock_ghz": "GRBM_GUI_ACTIVE / 8 / kernel duration of the same pass"},
           "kernels": {}}
    keep = ("gemm", "attn", "ln_", "vis_", "txt_", "pool", "rows_sum", "patchify", "splitk", "row_jobs", "clip_loss", "lse_rows", "align_", "cp_", "transpose2", "sgd_step")
    order = sorted(counters, key=lambda k: -counters[k].get("FETCH_SIZE", (0, 0.0))[1])
    for k in order:
        if not k.startswith(keep):
            continue
        cs = counters[k]
        e = {"launches": max(n for n, _ in cs.values())}
        per = lambda c: cs[c][1] / cs[c][0] if c in cs and cs[c][0] else None  # noqa: E731
        f, w = per("FETCH_SIZE"), per("WRITE_SIZE")
        if f is not None:
            e["fetch_kib_per_launch"] = round(f, 1)
        if w is not None:
            e["write_kib_per_launch"] = round(w, 1)
        if f is not None and w is not None:
            e["hbm_mb_per_launch"] = round((2 * f + w) * 1024 / 1e6, 2)
        g, mb = per("GRBM_GUI_ACTIVE"), per("SQ_VALU_MFMA_BUSY_CYCLES")
        if g and mb is not None:
            e["mfma_busy_frac"] = round(mb / (g / 8 * 1024), 4)
        if g and k in durs and durs[k][0]:
            avg_ns = durs[k][1] / durs[k][0]
            e["avg_us_in_counter_pass"] = round(avg_ns / 1e3, 1)
            e["clock_ghz"] = round(g / 8 / avg_ns, 3)
        for c in sorted(c for c in cs if c.startswith("SQ_")):
            v = per(c)
            if v is not None:
                e[c.lower() + "_per_launch"] = round(v, 1)
        if per("SQ_WAVE_CYCLES"):
            wc = per("SQ_WAVE_CYCLES")
            for c, name in (("SQ_WAIT_INST_ANY", "issue_stall_frac"), ("SQ_WAIT_ANY", "parked_frac"), ("SQ_ACTIVE_INST_ANY", "active_frac")):
                if per(c) is not None:
                    e[name] = round(per(c) / wc, 4)
        if per("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_frac"] = round((per("SQ_LDS_BANK_CONFLICT") or 0.0) / per("SQ_LDS_IDX_ACTIVE"), 4)
        out["kernels"][k] = e
    json.dump(out, open(out_path, "w"), indent=1)
    for k in ("gemm256p_kernel", "attn_bwd4_kernel", "attn_fwd_pair_kernel"):
        print(k, json.dumps(out["kernels"].get(k)))


if __name__ == "__main__":
    main()

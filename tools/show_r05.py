#!/usr/bin/env python3
"""One line per bench JSON: the headline and the round-5 sub-records.   python tools/show_r05.py <bench.json> [...]"""
import json
import sys

for f in sys.argv[1:]:
    j = json.loads(open(f).read().strip().splitlines()[-1])
    g = lambda k, kk="value": (j.get(k) or {}).get(kk)  # noqa: E731
    print(f, "| bf16", j["value"], j["ms_per_step"], "frac", j["step_mfma_frac"], "| f16", g("f16_mode"), "| f32", g("parity_mode"), "| fwd", g("fwd_only"), "| L/14", g("vit_l14"),
          "| plugin", g("plugin_step"), g("plugin_step", "vs_bare_step"), "| u8", g("plugin_step_u8"), g("plugin_step_u8", "vs_bare_step"),
          "| ref-order", g("plugin_step_reference_order"), "| gemm", j["roofline"]["achieved"], j["roofline"]["frac"], "traffic", j["roofline"]["traffic"],
          "| cpu", g("cpu_baseline"))

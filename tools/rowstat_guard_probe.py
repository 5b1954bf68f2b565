#!/usr/bin/env python3
"""How far do the one-sweep LayerNorm statistics (E[x^2] - mean^2 in f32) take the bf16 mode's features from the f32 HIP path when a vision block adds a
uniform offset to the residual stream (rows at |mean| = offset / std), and what does the guard's switch to the two-sweep statistics pass buy?
   python tools/rowstat_guard_probe.py [offset ...]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from lpi_amd import engine as E, synth  # noqa: E402
from lpi_amd.engine import DualEncoder, PackedIds  # noqa: E402
from lpi_amd.functional import DecomposedPromptFn  # noqa: E402

DEV = "cuda:0"
cfg = synth.CONFIGS["ViT-B/16"]
B = 32
img = torch.from_numpy(synth.images(B, 224)).to(DEV)
ids = synth.token_ids(B)
fac = synth.prompt_factors(9, 16, cfg.vision_width, cfg.transformer_width)
vis, txt = DecomposedPromptFn.apply(*[torch.from_numpy(fac[k]).to(DEV) for k in synth.PROMPT_NAMES])


def feats(enc, train=False):
    with torch.no_grad():
        (fi, cv), (ft, _) = enc.encode_both(img, PackedIds(ids).to(DEV), vis, txt, 3, train=train)
    torch.cuda.synchronize()
    return fi.clone(), ft.clone(), cv


def stat_err(enc):
    """max relative error of the ln_1 rstd the vision tower used in blocks 4.., against two-pass f64 statistics of the stored stream rows"""
    _, _, cv = feats(enc, train=True)
    ws = cv[0]
    M, worst, ratio = ws["M"], 0.0, 0.0
    for i in range(4, 11):
        x = ws["x"][i][:M].double()
        ref = 1.0 / (x.var(1, unbiased=False) + 1e-5).sqrt()
        got = ws["stat"][i][1][:M].double()
        worst = max(worst, float((got / ref - 1).abs().max()))
        ratio = max(ratio, float((x.mean(1).abs() / x.std(1, unbiased=False)).max()))
    return worst, ratio


for off in [float(a) for a in sys.argv[1:]] or [40.0, 150.0, 400.0, 1000.0]:
    sd = {k: np.array(v, copy=True) for k, v in synth.clip_state_dict(cfg).items()}
    sd["visual.transformer.resblocks.2.mlp.c_proj.bias"] += np.float32(off)
    enc32 = DualEncoder(cfg, sd, dtype="f32", device=DEV)
    ri, rt, _ = feats(enc32)
    del enc32
    out = {}
    for guard in (False, True):
        enc = DualEncoder(cfg, sd, dtype="bf16", device=DEV, options=E.EngineOptions(rowstat_guard=guard))
        for _ in range(3):
            fi, ft, _ = feats(enc)
        out[guard] = (float((fi - ri).abs().max()), stat_err(enc), enc.vis.rowstats, enc.rowstat_guard_tripped)
        del enc
    print(f"offset {off:7.1f}: guard off: feature err {out[False][0]:.3e}, rstd rel err {out[False][1][0]:.3e} (|mean|/std up to {out[False][1][1]:.0f}), rowstats {out[False][2]}"
          f" | guard on: feature err {out[True][0]:.3e}, rstd rel err {out[True][1][0]:.3e}, rowstats {out[True][2]}, rows counted {out[True][3]}", flush=True)

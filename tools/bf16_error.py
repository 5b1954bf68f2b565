#!/usr/bin/env python3
"""Report the bf16-mode error of the HIP path against the reference fixture (ViT-B/16, bs=8, depth 1) and the f32-mode error."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lpi_amd import synth
from lpi_amd.engine import DualEncoder
from lpi_amd.step import train_step
g = dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "vitb16_d1.npz")))
cfg = synth.VIT_B16
sd = synth.clip_state_dict(cfg)
for dtype in ("f32", "bf16"):
    enc = DualEncoder(cfg, sd, dtype=dtype, device="cuda:0")
    fac = {k: torch.from_numpy(v).to("cuda:0").requires_grad_(True) for k, v in synth.prompt_factors(9, 16, 768, 512).items()}
    out = train_step(enc, torch.from_numpy(synth.images(8, 224)).to("cuda:0"), torch.from_numpy(g["token_ids"]).to("cuda:0"), fac, 1)
    logits = (enc.logit_scale_exp * out["img_f"] @ out["txt_f"].t()).cpu().numpy()
    print(f"{dtype}: max|img_f err| {np.abs(out['img_f'].cpu().numpy() - g['img_f']).max():.2e}  max|txt_f err| {np.abs(out['txt_f'].cpu().numpy() - g['txt_f']).max():.2e}"
          f"  max|logit err| {np.abs(logits - g['logits']).max():.2e}  base_loss {float(out['base_loss']):.6f} (ref {float(g['base_loss']):.6f})")
    for k in synth.PROMPT_NAMES:
        a, b = fac[k].grad.cpu().numpy().ravel(), g["grad." + k].ravel()
        print(f"   grad {k:14s} rel max err {np.abs(a - b).max() / np.abs(b).max():.2e}  cosine {float(a @ b / np.sqrt((a @ a) * (b @ b))):.6f}")
    del enc

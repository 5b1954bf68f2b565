"""The text tower's SHARED-PREFIX layout (engine.PackedIds(shared=17), include/lpi_hip.h) against the plain packed layout: same features, same factor
gradients (up to the rounding of intermediate bf16 stores), and the step time of both, interleaved on one box.

    python tools/shared_prefix_probe.py [--model ViT-B/16] [--batch 256] [--dtype bf16] [--steps 30] [--rounds 3]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lpi_amd import synth  # noqa: E402
from lpi_amd.engine import DualEncoder, PackedIds  # noqa: E402
from lpi_amd.optim import flatten  # noqa: E402
from lpi_amd.step import train_step  # noqa: E402

DEV = torch.device("cuda:0")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="ViT-B/16")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--depth", type=int, default=3)
    a = ap.parse_args()
    cfg = synth.CONFIGS[a.model]
    enc = DualEncoder(cfg, synth.clip_state_dict(cfg), dtype=a.dtype, device=DEV)
    B = a.batch
    images = torch.from_numpy(synth.images(B, cfg.image_resolution)).to(DEV)
    ids_host = synth.token_ids(B)
    layouts = {"plain": PackedIds(ids_host).to(DEV), "shared": PackedIds(ids_host, shared=17).to(DEV)}
    print({k: v.rows for k, v in layouts.items()}, flush=True)
    fac = {k: torch.from_numpy(v).to(DEV).requires_grad_(True)
           for k, v in synth.prompt_factors(max(9, a.depth), 16, cfg.vision_width, cfg.transformer_width, r=4).items()}
    flat, flat_grad, views = flatten(fac)

    def step(ids):
        return train_step(enc, images, ids, fac, a.depth, None, flat_grad=flat_grad, grad_views=views)

    res = {}
    for name, ids in layouts.items():
        out = step(ids)
        torch.cuda.synchronize()
        res[name] = (out["img_f"].float().cpu().numpy().copy(), out["txt_f"].float().cpu().numpy().copy(), flat_grad.cpu().numpy().copy(),
                     {k: fac[k].grad.cpu().numpy().copy() for k in fac})
    p, s = res["plain"], res["shared"]
    print("image features max |diff|", float(np.abs(p[0] - s[0]).max()))
    print("text  features max |diff|", float(np.abs(p[1] - s[1]).max()), " (cosine min", float((p[1] * s[1]).sum(-1).min()), ")")
    for k in fac:
        d, m = float(np.abs(p[3][k] - s[3][k]).max()), float(np.abs(p[3][k]).max())
        print(f"grad {k:>16}: max |diff| {d:.3e}  max |ref| {m:.3e}  rel {d / max(m, 1e-30):.3e}", flush=True)

    for r in range(a.rounds):
        for name, ids in layouts.items():
            for _ in range(5):
                step(ids)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(a.steps):
                step(ids)
            e1.record()
            torch.cuda.synchronize()
            print(f"round {r} {name:>6}: {e0.elapsed_time(e1) / a.steps:.3f} ms / step", flush=True)


if __name__ == "__main__":
    main()

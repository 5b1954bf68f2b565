#!/usr/bin/env python3
"""The plugin's hot loop alone (SPrompts.train_epoch over host f32 images + caption strings, ViT-B/16, 256 pairs, bf16, depth 3) for a profiler:
   rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d <dir> -- python3 tools/plugin_loop.py [iterations] [u8|f32]"""
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from torch.utils.data import DataLoader  # noqa: E402
from lpi_amd.retrieval.methods.sprompt import SPrompts  # noqa: E402
from lpi_amd.retrieval.utils.data import SyntheticCoco, collate_keep_images  # noqa: E402
from lpi_amd.synth_bpe import ensure_vocab  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
pf = sys.argv[2] if len(sys.argv) > 2 else "f32"
ensure_vocab()
dev = torch.device("cuda:0")
args = json.load(open(os.path.join(REPO, "lpi_amd", "retrieval", "configs", "lpi", "coco_lpi.json")))
args.update(device=[dev], compute_dtype="bf16", honor_prompt_depth=True, prompt_depth=3, batch_size=256, epochs=1, num_workers=0, pixel_format=pf)
m = SPrompts(args)
m._network.update_fc(0)
ds = SyntheticCoco((n + 2) * 256, [0], 224, captions="strings", image_pool=256, pixel_format=pf)
loader = DataLoader(ds, batch_size=256, shuffle=False, num_workers=0, collate_fn=collate_keep_images)
opt, _ = m._setup_training()
import time  # noqa: E402
t = {}


hm = []


def on_step(i, b, o):
    if i > 7 and hasattr(b, "host_ms"):
        hm.append(b.host_ms)
    if i == 7:
        torch.cuda.synchronize()
        t[0] = time.perf_counter()
    if i == n - 1:
        torch.cuda.synchronize()
        t[1] = time.perf_counter()
        return True
    return False


m.train_epoch(loader, opt, 0, None, on_step)
if hm:
    print("producer ms per batch:", {k: round(sum(h[k] for h in hm) / len(hm), 3) for k in hm[0]})
print("done", n, "iterations;", round(1e3 * (t[1] - t[0]) / (n - 8), 3), "ms per iteration after 8 warm-up iterations")
print("device memory: allocated", round(torch.cuda.memory_allocated() / 2**30, 2), "GiB, peak", round(torch.cuda.max_memory_allocated() / 2**30, 2), "GiB, reserved",
      round(torch.cuda.memory_reserved() / 2**30, 2), "GiB;  text layout shared rows:", m._network._shared_rows())

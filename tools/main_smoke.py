#!/usr/bin/env python3
"""End-to-end smoke of the plugin's own entry (retrieval/main.py -> trainer.train) at ViT-B/16 size on synthetic data: two tasks, caption strings, uint8 or f32
pixels; prints the wall time per phase.   python tools/main_smoke.py [u8|f32]"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lpi_amd.retrieval import trainer  # noqa: E402
from lpi_amd.synth_bpe import ensure_vocab  # noqa: E402

ensure_vocab()
args = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lpi_amd", "retrieval", "configs", "lpi", "coco_lpi.json")))
args.update(num_tasks=2, epochs=2, batch_size=256, synthetic_train_size=2048, synthetic_eval_images_per_task=64, synthetic_captions="strings",
            synthetic_image_pool=256, honor_prompt_depth=True, pixel_format=(sys.argv[1] if len(sys.argv) > 1 else "f32"), seed=[1993], device=["0"])
os.chdir(tempfile.mkdtemp())
t0 = time.perf_counter()
model = trainer._train(args)
print("seconds", round(time.perf_counter() - t0, 1), "final_res keys", list(model.final_res), "task 1 t2i", model.final_res[1]["mscoco"]["t2i"])

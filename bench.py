#!/usr/bin/env python3
"""bench.py — image-text pairs/s, forward+backward, of the LPI retrieval hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

One "step" = one pass of the hot path over one batch of synthetic input already resident in HBM
(SURVEY.md section 8(d)): DecomposedPrompt reconstruction -> prompted CLIP ViT-B/16 + text transformer forward ->
contrastive + alignment losses -> dgrad backward through all 24 blocks to the prompt slots -> CP-factor gradients
(+ RCCL all-gather of embeddings / all-reduce of factor grads when N > 1) -> SGD update of the 5 284 prompt parameters
(methods/sprompt.py:297-311).  Workload at N=1 = BASELINE.json configs[2]: ViT-B/16, bs=256/GPU, prompt_depth=3, r=4.

The JSON line carries, besides the driver's contract keys:
  roofline     : the dominant kernel (gemm_nt_kernel, MFMA bound): algorithmic FLOPs (2*M*N*K over un-padded M) of every GEMM
                 launch of a step / its duration, bracketed by HIP events on the launch stream in an instrumented pass that
                 follows the timed region; peak = dense MFMA peak of the operand dtype (bf16 2500 TF, f32 157.3 TF).
  step_mfma_frac: whole-step fraction of the same peak from SURVEY's 89.68 GFLOP/pair (all kernels, not just GEMMs).
  cpu_baseline : the oracle (oracle/lpi_oracle.py, "port") timed on this host's cores on a bounded bs=8 sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

GFLOP_PER_PAIR = {"ViT-B/16": (89.68, 44.05), "ViT-L/14": (378.9, 185.8)}   # (fwd+bwd, fwd), P=16, dgrad only
PEAK_TF = {"bf16": 2500.0, "f32": 157.3}   # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md:41-43


def cpu_baseline(cfg, depth, seconds_budget=25.0):
    """Oracle fwd+loss+bwd at bs=8 (BASELINE.json configs[0] shape) on the host cores."""
    from lpi_amd import synth
    from oracle import lpi_oracle as O

    B = 8
    orc = O.Oracle(cfg, synth.clip_state_dict(cfg))
    img = synth.images(B, cfg.image_resolution)
    ids = synth.token_ids(B)
    fac = synth.prompt_factors(9, 16, cfg.vision_width, cfg.transformer_width)
    O.train_step(orc, img, ids, fac, depth=depth)          # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        O.train_step(orc, img, ids, fac, depth=depth)
        n += 1
        el = time.perf_counter() - t0
        if el > seconds_budget or n >= 5:
            break
    return {"value": round(B * n / el, 3), "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} steps of bs={B} fwd+bwd, ViT-B/16 depth={depth} r=4, fp32, oracle/lpi_oracle.py on torch CPU "
                      f"({torch.get_num_threads()} threads of {os.cpu_count()} cpus)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--depth", type=int, default=3)
    ap.add_argument("--model", default="ViT-B/16")
    ap.add_argument("--rank", type=int, default=4, help="CP rank r of the DecomposedPrompt")
    ap.add_argument("--prompt-layers", type=int, default=9, help="layer_num of the DecomposedPrompt (reference: 9)")
    ap.add_argument("--fwd-only", action="store_true", help="BASELINE.json configs[1]: encoder forward + cosine matrix")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--overlap", action="store_true",
                    help="run the two towers on two HIP streams (was +6.6 %% with the first kernels; -0.7 %% with the final ones: A/B switch)")
    ap.add_argument("--no-overlap", action="store_true", help="accepted for older command lines: one stream is the default")
    ap.add_argument("--no-text-trim", action="store_true", help="compute all 77 text positions, also those behind every caption's EOT (A/B switch)")
    ap.add_argument("--vision-lanes", type=int, default=1, help="micro-batches of the vision tower on separate streams (measured null on MI355X)")
    ap.add_argument("--text-lanes", type=int, default=1)
    a = ap.parse_args()

    import torch.distributed as dist
    from lpi_amd import engine, synth
    from lpi_amd.engine import DualEncoder
    from lpi_amd.step import forward_loss, train_step

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    exchange = None
    if world > 1 or os.environ.get("LPI_FORCE_DIST") == "1":       # LPI_FORCE_DIST: exercise the RCCL path on a 1-GPU box
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)
        from lpi_amd.dp import Exchange
        exchange = Exchange()

    cfg = synth.CONFIGS[a.model]
    enc = DualEncoder(cfg, synth.clip_state_dict(cfg), dtype=a.dtype, device=dev)
    B = a.batch
    images = torch.from_numpy(synth.images(B, cfg.image_resolution, seed=synth.IMAGE_SEED + rank)).to(dev)
    ids_host = synth.token_ids(B, seed=synth.TOKEN_SEED + rank)          # [B, 77] as the tokenizer builds them, on the host
    if not a.no_text_trim:
        # token columns behind the longest caption's EOT are dead under the causal mask (engine.trim_token_ids): not computed
        from lpi_amd.engine import trim_token_ids
        ids_host = np.ascontiguousarray(trim_token_ids(ids_host))
    ids = torch.from_numpy(ids_host).to(dev)
    fac = {k: torch.from_numpy(v).to(dev).requires_grad_(not a.fwd_only)
           for k, v in synth.prompt_factors(max(a.prompt_layers, a.depth), 16, cfg.vision_width, cfg.transformer_width, r=a.rank).items()}
    opt = torch.optim.SGD(list(fac.values()), momentum=0.9, lr=0.05, weight_decay=2e-4)    # sprompt.py:253

    def step():
        if a.fwd_only:
            with torch.no_grad():
                forward_loss(enc, images, ids, fac, a.depth, exchange.gather if exchange else None, overlap_towers=a.overlap, vision_lanes=a.vision_lanes, text_lanes=a.text_lanes)
        else:
            train_step(enc, images, ids, fac, a.depth, exchange, overlap_towers=a.overlap, vision_lanes=a.vision_lanes, text_lanes=a.text_lanes)
            opt.step()

    def sync():
        if exchange is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync()
    el = time.perf_counter() - t0
    if exchange is not None:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    pairs_s = world * B * a.steps / el

    roofline = None
    if not a.no_roofline:
        engine.GEMM_PROFILE = []
        nprof = 2
        overlap_saved, a.overlap = a.overlap, False     # time each kernel alone: no second stream sharing the GPU
        for _ in range(nprof):
            step()
        a.overlap = overlap_saved
        torch.cuda.synchronize()
        ev_all = engine.GEMM_PROFILE
        engine.GEMM_PROFILE = None
        # the dominant kernel = the 256x256 GEMM; the few-row GEMMs (split-K 128x128 + reduce) and the half-empty launches that go to
        # the 256x128-tile kernel are reported beside it, not averaged into its launch time
        ev = [e for e in ev_all if e[4] == "k256"] or ev_all
        small = [e for e in ev_all if e[4] == "few_rows"]
        small_ms = sum(e[0].elapsed_time(e[1]) for e in small)
        half = [e for e in ev_all if e[4] == "k256x128"]
        half_ms = sum(e[0].elapsed_time(e[1]) for e in half)
        half_fl = sum(e[2] for e in half)
        ms = sum(e[0].elapsed_time(e[1]) for e in ev)
        fl = sum(e[2] for e in ev)
        by = sum(e[3] for e in ev)
        ach = fl / (ms * 1e-3) / 1e12
        traffic, tsrc = None, None
        pmc = os.path.join(REPO, "profiles", "r01_pmc_traffic.json")
        if a.dtype == "bf16" and os.path.exists(pmc):      # measured by separate rocprofv3 --pmc passes (tools/pmc_summary.py)
            ks = [v for n, v in json.load(open(pmc))["kernels"].items() if n in ("gemm256_kernel", "gemm256_tail_kernel")]
            if ks:      # the two entry kernels of the 256x256 GEMM, launch-weighted
                traffic = round(sum(v["hbm_mb_per_launch"] * v["launches"] for v in ks) / sum(v["launches"] for v in ks) * 1e6)
                tsrc = "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, corrected)"
        roofline = {"bound": "mfma", "kernel": "gemm256_kernel / gemm256_tail_kernel (the 256x256 GEMM; the second runs a short last round as half tiles)",
                    "achieved": round(ach, 2),
                    "peak": PEAK_TF[a.dtype], "unit": "TFLOP/s", "frac": round(ach / PEAK_TF[a.dtype], 4),
                    "traffic": traffic, "traffic_unit": "HBM bytes per launch", "traffic_source": tsrc,
                    "algorithmic_bytes_per_launch": round(by / len(ev)),
                    "launches_per_step": len(ev) // nprof, "avg_launch_us": round(1e3 * ms / len(ev), 2),
                    "gemm_ms_per_step": round(ms / nprof, 3), "gemm_gflop_per_step": round(fl / nprof / 1e9, 1),
                    "few_row_gemms": {"launches_per_step": len(small) // nprof, "ms_per_step": round(small_ms / nprof, 3),
                                      "kernel": "gemm_nt_kernel split-K + splitk_reduce_kernel"},
                    "half_empty_gemms": {"launches_per_step": len(half) // nprof, "ms_per_step": round(half_ms / nprof, 3),
                                         "achieved_tflops": round(half_fl / (half_ms * 1e-3) / 1e12, 2) if half_ms else None,
                                         "kernel": "gemm256x128_kernel (launches with 16..159 256x256 tiles)"},
                    "measured": "HIP events around every GEMM launch of 2 extra steps, towers on one stream (kernel alone on the GPU)"}

    if rank == 0:
        gfs = GFLOP_PER_PAIR.get(a.model)
        gf = None if gfs is None else gfs[1 if a.fwd_only else 0]
        out = {
            "metric": (f"image-text pairs/sec fwd+bwd ({a.model}, bs{B}/GPU)" if not a.fwd_only
                       else f"image-text pairs/sec fwd-only encoder + cosine matrix ({a.model}, bs{B}/GPU)"),
            "value": round(pairs_s, 2), "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * el / a.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": ("BASELINE.json configs[2]: " if (a.model == "ViT-B/16" and B == 256 and a.depth == 3 and a.rank == 4 and not a.fwd_only) else "")
                       + f"{a.model} dual encoder bs={B}/GPU prompt_depth={a.depth} r={a.rank} P=16, "
                       + ("fwd-only + cosine matrix" if a.fwd_only else "fwd+bwd incl. DecomposedPrompt grads + SGD step"),
                       "global_batch": world * B, "image": f"{cfg.image_resolution}x{cfg.image_resolution}", "tokens": cfg.context_length,
                       "text_rows_computed": int(ids.shape[1]),   # < tokens: columns behind the longest caption's EOT are dead (causal mask) and skipped, exactly
                       "parallelism": f"dp{world}", "weights": "synthetic (numpy Philox, CLIP-init scales), frozen"},
            "step_mfma_frac": None if gf is None else round(pairs_s / world * gf * 1e9 / (PEAK_TF[a.dtype] * 1e12), 4),
            "roofline": roofline,
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, a.depth)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if exchange is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

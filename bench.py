#!/usr/bin/env python3
"""bench.py — image-text pairs/s, forward+backward, of the LPI retrieval hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 works as typed: when the process was not started by torch.distributed.run (no WORLD_SIZE in the environment) it starts
`python -m torch.distributed.run --nproc-per-node N ... bench.py <same arguments>` as a CHILD process before anything touches the
GPU, relays rank 0's JSON line and exits with the child's code.  Launched by torch.distributed.run (the driver's way), each rank
reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.  One rank per GPU over RCCL (backend "nccl").

One "step" = one pass of the hot path over one batch of synthetic input already resident in HBM
(SURVEY.md section 8(d)): DecomposedPrompt reconstruction -> prompted CLIP ViT-B/16 + text transformer forward ->
contrastive + alignment losses -> dgrad backward through all 24 blocks to the prompt slots -> CP-factor gradients
(+ RCCL all-gather of embeddings / all-reduce of factor grads when N > 1) -> SGD update of the 5 284 prompt parameters
(methods/sprompt.py:297-311).  Workload at N=1 = BASELINE.json configs[2]: ViT-B/16, bs=256/GPU, prompt_depth=3, r=4.

The JSON line carries, besides the driver's contract keys:
  value / ms_per_step : from the wall clock around EXACTLY K steps (barrier + synchronize on both sides, MAX over ranks);
  median_ms_per_step  : median of the K per-step durations (HIP events recorded after every step inside that same region);
  roofline     : the dominant kernel (the 256x256 GEMM, MFMA bound): executed FLOPs of its launches / their duration, bracketed by
                 HIP events on the launch stream in an instrumented pass after the timed region; which kernel a launch went to is
                 reported by the library (lpi_gemm_last_kernel), not re-derived here; each launch counts with the
                 median of five instrumented steps; `traffic` / `mfma_util` come from a committed rocprofv3 --pmc summary
                 (profiles/r*_pmc*.json) of the SAME workload (model / batch / depth), dtype, tuning and library build — else null;
  step_mfma_frac: whole-step fraction of the same peak from SURVEY's 89.68 GFLOP/pair — model-FLOP utilisation (all kernels, not just GEMMs);
  hw_flop_frac : the same with the FLOPs the kernels actually execute (dead text rows, the pooled last block, the prompt-row backward of the
                 first block are skipped exactly, so this is lower);
  parity_mode  : (N = 1) the f32-operand mode that meets the 1e-4 parity bar, same workload, 20 steps, its own roofline block (157.3 TF peak);
  vit_l14      : (N = 1) BASELINE.json configs[4]'s per-GPU workload (ViT-L/14, 512 pairs, depth 12, r 8, bf16);
  fwd_only     : (N = 1) BASELINE.json configs[1]: encoder forward + cosine matrix;
  eval_path    : (N = 1) the reference's evaluation path per batch (task-id pass, L1 task selection, per-sample prompted forward), images/s and
                 captions/s, and the score matrix + ranks at COCO 5k-test size;
  f16_mode     : (N = 1) the same step with fp16 MFMA operands in the forward (the reference's arithmetic type): 4x lower logit error;
  parity       : (N = 1) what the headline's arithmetic is worth, measured in this very run: the f32 HIP step against the fixture captured from the imported
                 reference (tests/golden/vitb16_d3_patched.npz: ViT-B/16, 8 pairs, depth 3 — a fixture is data, not the oracle) and the bf16 / f16 steps
                 against the f32 HIP step on the benchmarked batch (256 pairs); the 1e-4 bar belongs to the first, the throughput modes are held to the second;
  plugin_step  : (N = 1) the reference's real caller loop — SPrompts.train_epoch (methods/sprompt.py:290-334) built from configs/lpi/coco_lpi.json, fed by a
                 DataLoader over a synthetic Coco that yields HOST f32 images and caption STRINGS: H2D copy, tokenisation, forward, losses, backward, SGD
                 step; pairs/s from the wall clock around K loop iterations, its ratio to the bare step (`value`) and the split of a step;
  collectives  : (N > 1) mean microseconds of the feature all-gather and the factor-gradient all-reduce per step;
  cpu_baseline : the oracle (oracle/lpi_oracle.py, "port") timed on this host's cores on a bounded bs=8 sample, thread count swept.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

GFLOP_PER_PAIR = {"ViT-B/16": (89.68, 44.05), "ViT-L/14": (378.9, 185.8)}   # (fwd+bwd, fwd), P=16, dgrad only
PEAK_TF = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}   # dense MFMA peaks, /opt/skills/guides/MI355X_MICROARCH.md:41-43
GEMM_KERNEL_NAMES = {0: "k128", 1: "k256", 2: "k256", 3: "k256x128", 4: "few_rows"}     # LPI_GEMM_K_* -> report bucket


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--depth", type=int, default=3)
    ap.add_argument("--model", default="ViT-B/16")
    ap.add_argument("--rank", type=int, default=4, help="CP rank r of the DecomposedPrompt")
    ap.add_argument("--prompt-layers", type=int, default=9, help="layer_num of the DecomposedPrompt (reference: 9)")
    ap.add_argument("--fwd-only", action="store_true", help="BASELINE.json configs[1]: encoder forward + cosine matrix")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the parity_mode / fwd_only sub-records")
    ap.add_argument("--overlap", action="store_true",
                    help="run the two towers on two HIP streams (was +6.6 %% with the first kernels; -0.7 %% with the final ones: A/B switch)")
    ap.add_argument("--no-overlap", action="store_true", help="accepted for older command lines: one stream is the default")
    ap.add_argument("--no-lockstep", action="store_true",
                    help="run the towers one after the other instead of in lock step with grouped GEMM launches (A/B switch; same bits)")
    ap.add_argument("--no-text-pack", action="store_true",
                    help="cut the text batch at the LONGEST caption only instead of packing every caption at its own length (A/B switch)")
    ap.add_argument("--no-text-shared", action="store_true",
                    help="store SOT and the 16 context slots per sample instead of ONCE for the batch (engine.PackedIds(shared=17): the prompts are broadcast, so "
                         "under the causal mask those 17 positions hold the same rows for every sample in every block; A/B switch)")
    ap.add_argument("--no-text-trim", action="store_true", help="compute all 77 text positions, also those behind every caption's EOT (A/B switch)")
    ap.add_argument("--vision-lanes", type=int, default=1, help="micro-batches of the vision tower on separate streams (measured null on MI355X)")
    ap.add_argument("--text-lanes", type=int, default=1)
    ap.add_argument("--local-loss", action="store_true",
                    help="N > 1: the row-sharded contrastive loss (gather_features local_loss=True, sprompt.py:278-283): a rank evaluates its B x W B logit "
                         "blocks only, labels offset by rank * B")
    ap.add_argument("--gather-with-grad", action="store_true",
                    help="N > 1: gradients through the gathered features (sprompt.py:67-69): the key gradients are reduce-scattered to their owners")
    ap.add_argument("--engine-opt", action="append", default=[], metavar="FIELD=VALUE",
                    help="an lpi_amd.engine.EngineOptions field for every engine this run builds (A/B switch), e.g. --engine-opt stream_pool=0 --engine-opt ln_fold=1")
    ap.add_argument("--no-affinity", action="store_true", help="N > 1: leave the ranks' CPU affinity alone (default: every rank gets its own slice of its GPU's NUMA node)")
    ap.add_argument("--force-dist", action="store_true", help="N = 1: run the step through a one-rank RCCL process group (the code path an 8-GPU run takes)")
    ap.add_argument("--cpu-baseline-bs256", action="store_true", help="also time ONE bs=256 oracle step on the host (~70 s; needs 200 GiB of host memory and 32 CPUs)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="N > 1 ranks on ONE GPU with a gloo group (messages staged through the host): exercises the multi-rank path on a 1-GPU box")
    return ap.parse_args()


def engine_options(a):
    """EngineOptions from the environment fall-backs and the --engine-opt FIELD=VALUE arguments (None when there are none: the engine's own default)."""
    if not a.engine_opt:
        return None
    from lpi_amd.engine import EngineOptions
    kw = {}
    for item in a.engine_opt:
        k, _, v = item.partition("=")
        cur = getattr(EngineOptions(), k)      # AttributeError for an unknown field
        kw[k] = (v not in ("0", "false", "False", "")) if isinstance(cur, bool) else int(v)
    return EngineOptions.from_env(**kw)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _cpulist(text):
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def gpu_numa_nodes(sysfs="/sys"):
    """NUMA node of every GPU in HIP's device order, read from sysfs WITHOUT touching the GPU: the KFD topology lists the nodes in the order ROCr enumerates
    them (CPU nodes have simd_count 0), a GPU node's `drm_render_minor` names its DRM render node, and /sys/class/drm/renderD<minor>/device/numa_node is the
    PCI device's NUMA node (-1 where the platform does not say).  Fallback: the amdgpu cards of /sys/class/drm in PCI bus order.  HIP_VISIBLE_DEVICES /
    ROCR_VISIBLE_DEVICES given as index lists are applied.  -> list of ints (-1 = unknown), or None when sysfs has nothing to say."""
    import glob

    def read(path):
        try:
            with open(path) as f:
                return f.read().strip()
        except OSError:
            return None
    nodes = []
    kfd = sorted(glob.glob(os.path.join(sysfs, "class/kfd/kfd/topology/nodes/[0-9]*")), key=lambda q: int(os.path.basename(q)))
    for d in kfd:
        props = dict(ln.split(None, 1) for ln in (read(os.path.join(d, "properties")) or "").splitlines() if " " in ln)
        if int(props.get("simd_count", "0") or 0) <= 0:
            continue
        minor = props.get("drm_render_minor")
        nn = read(os.path.join(sysfs, f"class/drm/renderD{minor}/device/numa_node")) if minor else None
        nodes.append(int(nn) if nn not in (None, "") else -1)
    if not nodes:
        cards = []
        for c in glob.glob(os.path.join(sysfs, "class/drm/card[0-9]*")):
            if not os.path.basename(c)[4:].isdigit() or read(os.path.join(c, "device/vendor")) != "0x1002":
                continue
            nn = read(os.path.join(c, "device/numa_node"))
            cards.append((os.path.basename(os.path.realpath(os.path.join(c, "device"))), int(nn) if nn not in (None, "") else -1))
        nodes = [n for _, n in sorted(cards)]
    if not nodes:
        return None
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES"):      # ROCr filters first, HIP filters what is left
        v = os.environ.get(var)
        if v and all(t.strip().isdigit() for t in v.split(",")):
            idx = [int(t) for t in v.split(",")]
            if all(i < len(nodes) for i in idx):
                nodes = [nodes[i] for i in idx]
    return nodes


def pin_rank_to_cpus(local_rank, local_world, share_gpu=False, enabled=True):
    """Give this rank its own slice of the host CPUs BEFORE anything touches the GPU: W processes that each spawn library / pipeline threads on all cores
    migrate across sockets and stall each other's launch loops.  The allowed CPUs are split by NUMA node first — local rank r uses GPU r, whose node comes
    from sysfs (gpu_numa_nodes: KFD topology -> DRM render node -> PCI numa_node; round 6: this replaced the guess that device order follows the sockets,
    which stays the fallback where the platform reports no node) — then evenly among the ranks that share a node.  enabled = False
    (bench.py --no-affinity) leaves the affinity alone.  Returns the CPU list (or None)."""
    if not enabled or local_world <= 1 or not hasattr(os, "sched_setaffinity"):
        return None
    allowed = sorted(os.sched_getaffinity(0))
    nodes = []
    try:
        import glob
        for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*"), key=lambda q: int(q.rsplit("node", 1)[1])):
            cpus = [c for c in _cpulist(open(os.path.join(d, "cpulist")).read()) if c in set(allowed)]
            if cpus:
                nodes.append(cpus)
    except OSError:
        nodes = []
    if not nodes:
        nodes = [allowed]
    # which NUMA node a local rank's GPU hangs off: GPU g of the host's n GPUs -> node g * nodes / n (device order follows the sockets on the MI355X
    # platforms); rank r uses GPU r.  The ranks that share a node split its CPUs evenly.
    try:
        import torch
        ngpu = max(int(torch.cuda.device_count()), local_world)      # does not initialise the GPU on this image
    except Exception:      # noqa: BLE001
        ngpu = local_world
    node_of = [min(len(nodes) - 1, r * len(nodes) // ngpu) for r in range(local_world)]      # fallback: device order follows the sockets
    node_ids = []
    try:
        import glob as _g
        node_ids = sorted(int(q.rsplit("node", 1)[1]) for q in _g.glob("/sys/devices/system/node/node[0-9]*")
                          if any(c in set(allowed) for c in _cpulist(open(os.path.join(q, "cpulist")).read())))
    except (OSError, ValueError):
        node_ids = []
    gn = gpu_numa_nodes()
    gpu_of = (lambda r: 0) if share_gpu else (lambda r: r)      # --share-gpu: every rank runs on GPU 0
    if gn is not None and len(gn) > max(gpu_of(r) for r in range(local_world)) and len(node_ids) == len(nodes) \
            and all(gn[gpu_of(r)] in node_ids for r in range(local_world)):
        node_of = [node_ids.index(gn[gpu_of(r)]) for r in range(local_world)]
    mates = [r for r in range(local_world) if node_of[r] == node_of[local_rank]]
    cpus = nodes[node_of[local_rank]]
    k = mates.index(local_rank)
    share = max(1, len(cpus) // len(mates))
    mine = cpus[k * share:(k + 1) * share] or cpus
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    return mine


def self_launch(a):
    """`python bench.py --gpus N` typed by hand: start the ranks as a child torch.distributed.run BEFORE any GPU call in this process
    (a process that has initialised the GPU must never exec or be replaced), relay rank 0's JSON line, return the child's exit code."""
    import torch
    ndev = torch.cuda.device_count()          # does not initialise the GPU on this image
    if a.gpus > ndev and not a.share_gpu:
        print(f"bench.py: --gpus {a.gpus} but only {ndev} GPU(s) visible (use --share-gpu to run the ranks on one GPU over gloo)", file=sys.stderr)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if p.returncode != 0 or line is None:
        print(f"bench.py: the {a.gpus}-rank child run failed (exit code {p.returncode})", file=sys.stderr)
        return p.returncode or 1
    print(line, flush=True)
    return 0


def cpu_baseline(cfg, depth, seconds_budget=30.0, bs256=False):
    """Oracle fwd+loss+bwd at bs=8 (BASELINE.json configs[0] shape) on the host cores.  torch's default (all logical CPUs) oversubscribes
    a bs=8 step, so the thread count is swept first (one step each, after a warm-up) and the best one is timed: >= 3 warm-up steps, then
    >= 10 timed steps, the MEDIAN step time is reported (the mean beside it).  When the host has the memory (the oracle keeps
    ~0.3 GB of autograd state per pair) one bs=256 step — the metric's own batch — is timed as well."""
    import numpy as np
    import torch
    from lpi_amd import synth
    from oracle import lpi_oracle as O

    B = 8
    orc = O.Oracle(cfg, synth.clip_state_dict(cfg))
    img = synth.images(B, cfg.image_resolution)
    ids = synth.token_ids(B)
    fac = synth.prompt_factors(9, 16, cfg.vision_width, cfg.transformer_width)
    ncpu = os.cpu_count() or 1
    default_threads = torch.get_num_threads()
    cand = sorted({t for t in (8, 16, 32, 64) if t <= ncpu} | {min(ncpu, default_threads)})
    t_start = time.perf_counter()
    sweep = {}
    for t in cand:
        torch.set_num_threads(t)
        if not sweep:
            O.train_step(orc, img, ids, fac, depth=depth)          # warm-up (allocator, first-touch)
        t0 = time.perf_counter()
        O.train_step(orc, img, ids, fac, depth=depth)
        sweep[t] = time.perf_counter() - t0
        if time.perf_counter() - t_start > 0.5 * seconds_budget:
            break
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    for _ in range(3):                                              # warm-up at the chosen thread count
        O.train_step(orc, img, ids, fac, depth=depth)
    per, t_loop = [], time.perf_counter()
    while len(per) < 10 or (time.perf_counter() - t_loop < 0.25 * seconds_budget and len(per) < 40):
        t0 = time.perf_counter()
        O.train_step(orc, img, ids, fac, depth=depth)
        per.append(time.perf_counter() - t0)
    med, mean = float(np.median(per)), float(np.mean(per))
    out = {"value": round(B / med, 3), "unit": "pairs/s", "cores": best, "kind": "port", "value_at_mean": round(B / mean, 3),
           "timed_steps": len(per), "warmup_steps": 3,
           "sample": f"median of {len(per)} timed steps (3 warm-up) of bs={B} fwd+bwd, ViT-B/16 depth={depth} r=4, fp32, oracle/lpi_oracle.py on torch CPU with "
                     f"{best} threads (best of a one-step sweep {{{', '.join(f'{t}: {B / v:.2f}' for t, v in sweep.items())}}} pairs/s; {ncpu} logical cpus)"}
    # the metric's own batch: one bs=256 step, when the host can hold it (~80 GB of autograd state; 64 threads: the large batch scales further)
    try:
        import psutil
        avail = psutil.virtual_memory().available / 2**30
    except Exception:
        avail = 0.0
    # opt-in since round 5 (LPI_CPU_BASELINE_BS256=1): the one un-warmed bs=256 step takes 70 s of host time — beyond the 10-30 s a bounded sample should cost
    if avail >= 200.0 and ncpu >= 32 and bs256:
        Bl, tl = 256, min(64, ncpu)
        torch.set_num_threads(tl)
        imgl, idsl = synth.images(Bl, cfg.image_resolution), synth.token_ids(Bl)
        t0 = time.perf_counter()
        O.train_step(orc, imgl, idsl, fac, depth=depth)
        el = time.perf_counter() - t0
        out["bs256_sample"] = {"value": round(Bl / el, 3), "unit": "pairs/s", "cores": tl, "steps": 1, "seconds": round(el, 2),
                               "sample": f"ONE bs={Bl} step (no warm-up at this size), {tl} threads; host memory available {avail:.0f} GiB"}
        del imgl, idsl
    else:
        out["bs256_sample"] = None
        out["bs256_skipped"] = ("opt-in (bench.py --cpu-baseline-bs256): one bs=256 step costs ~70 s of host time; round 4 measured 3.5 pairs/s at 64 threads"
                                if not bs256 else f"host memory available {avail:.0f} GiB (< 200) or {ncpu} cpus (< 32)")
    torch.set_num_threads(default_threads)
    return out


class Workload:
    """Engine + synthetic inputs + optimiser for one (dtype, fwd_only) configuration on this rank."""

    def __init__(self, a, dev, rank, dtype, fwd_only, exchange):
        import numpy as np
        import torch
        from lpi_amd import synth
        from lpi_amd.engine import DualEncoder, PackedIds, trim_token_ids
        self.a, self.dev, self.dtype, self.fwd_only, self.exchange = a, dev, dtype, fwd_only, exchange
        self.cfg = cfg = synth.CONFIGS[a.model]
        self.enc = DualEncoder(cfg, synth.clip_state_dict(cfg), dtype=dtype, device=dev, options=engine_options(a))
        B = a.batch
        self.images = torch.from_numpy(synth.images(B, cfg.image_resolution, seed=synth.IMAGE_SEED + rank)).to(dev)
        ids_host = synth.token_ids(B, seed=synth.TOKEN_SEED + rank)          # [B, 77] as the tokenizer builds them, on the host
        if not a.no_text_trim:
            # token columns behind the longest caption's EOT are dead under the causal mask (engine.trim_token_ids): not computed
            ids_host = np.ascontiguousarray(trim_token_ids(ids_host))
        self.ids = torch.from_numpy(ids_host).to(dev)
        self.text_rows = float(self.ids.shape[1])
        self.text_shared = 0
        if not (a.no_text_trim or a.no_text_pack):
            # ... and so are the rows behind every caption's OWN EOT (engine.PackedIds: the text batch packed, one row per live token)
            # ... and the 17 positions every caption has in common (SOT + the broadcast context slots) are stored and computed once
            self.text_shared = 0 if a.no_text_shared else 17
            self.ids = PackedIds(ids_host, self.text_shared).to(dev)
            self.text_rows = self.ids.rows / B
        self.fac = {k: torch.from_numpy(v).to(dev).requires_grad_(not fwd_only)
                    for k, v in synth.prompt_factors(max(a.prompt_layers, a.depth), 16, cfg.vision_width, cfg.transformer_width, r=a.rank).items()}
        # sprompt.py:253: SGD(momentum 0.9, lr 0.05, weight decay 2e-4) — the factors live in one flat vector, their gradients in another:
        # the optimiser step is ONE lpi_sgd_step launch and the data-parallel all-reduce takes the flat gradient as it is
        from lpi_amd.optim import FlatSGD, flatten
        self.flat, self.flat_grad, self.grad_views = flatten(self.fac)
        self.opt = FlatSGD(self.fac, lr=0.05, momentum=0.9, weight_decay=2e-4, flat=self.flat, flat_grad=self.flat_grad, grad_views=self.grad_views)
        self.overlap = a.overlap

    def step(self):
        import torch
        from lpi_amd.step import forward_loss, train_step
        a = self.a
        if self.fwd_only:
            with torch.no_grad():
                forward_loss(self.enc, self.images, self.ids, self.fac, a.depth, self.exchange.gather if self.exchange else None,
                             overlap_towers=self.overlap, vision_lanes=a.vision_lanes, text_lanes=a.text_lanes, lockstep=not a.no_lockstep)
        else:
            train_step(self.enc, self.images, self.ids, self.fac, a.depth, self.exchange, overlap_towers=self.overlap,
                       vision_lanes=a.vision_lanes, text_lanes=a.text_lanes, lockstep=not a.no_lockstep,
                       flat_grad=self.flat_grad, grad_views=self.grad_views)
            self.opt.step()

    def run(self, steps, warmup, sync):
        """-> (seconds for exactly `steps` steps, per-step milliseconds from HIP events)."""
        import torch
        for _ in range(warmup):
            self.step()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        sync()
        t0 = time.perf_counter()
        ev[0].record()
        for i in range(steps):
            self.step()
            ev[i + 1].record()
        sync()
        el = time.perf_counter() - t0
        per = [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]
        return el, per

    def gemm_roofline(self, nprof=5):
        """Instrumented pass: HIP events around every GEMM launch; the library says which kernel each launch used.  An event pair also spans any time
        the GPU waits for a late host (the host creates two events per launch here: a cold host on a fresh box once inflated the mean launch
        2.5x), so one instrumented warm-up step runs first and every launch of the step counts with the MEDIAN of its nprof measurements —
        robust against a late host without being a best case (the fastest-of-n figure is reported beside it)."""
        import numpy as np
        import torch
        from lpi_amd import _lib, engine
        a = self.a
        overlap_saved, self.overlap = self.overlap, False     # time each kernel alone: no second stream sharing the GPU
        engine.GEMM_PROFILE = []
        self.step()                                           # instrumented warm-up, discarded
        torch.cuda.synchronize()
        engine.GEMM_PROFILE = []
        for _ in range(nprof):
            self.step()
        self.overlap = overlap_saved
        torch.cuda.synchronize()
        ev_raw = engine.GEMM_PROFILE
        engine.GEMM_PROFILE = None
        # every step issues the same launches in the same order: launch i of the step = the median of its nprof measurements
        nrec, per = nprof, len(ev_raw) // nprof
        same = per * nprof == len(ev_raw) and all(ev_raw[i][2:] == ev_raw[i + k * per][2:] for i in range(per) for k in range(1, nprof))
        if same:
            ev_all = []
            for i in range(per):
                ts = sorted(ev_raw[i + k * per][0].elapsed_time(ev_raw[i + k * per][1]) for k in range(nprof))
                ev_all.append((float(np.median(ts)), ts[0]) + tuple(ev_raw[i][2:]))
            nprof = 1
        else:
            ev_all = [(e[0].elapsed_time(e[1]),) * 2 + tuple(e[2:]) for e in ev_raw]
        bucket = lambda e: GEMM_KERNEL_NAMES.get(e[4], "other")  # noqa: E731
        # the DOMINANT kernel = the GEMM kernel with the most time in the step: the 256x256 kernel in the bf16 / f16 modes; in the f32 mode most launches
        # have fewer tiles than its threshold and run the 128x128 kernel (gemm_nt_kernel), which then IS the dominant one and is reported as such.
        # The other kernels are listed beside it, never averaged into its launch time.
        t_ms = lambda es: sum(e[0] for e in es)  # noqa: E731
        groups = {b: [e for e in ev_all if bucket(e) == b] for b in ("k256", "k128", "k256x128", "few_rows")}
        dom = max(groups, key=lambda b: t_ms(groups[b]))
        ev = groups[dom] or ev_all
        ms = t_ms(ev)
        ms_best = sum(e[1] for e in ev)
        self.gemm_gflop_all = sum(e[2] for e in ev_all) / nprof / 1e9          # every GEMM launch of a step (executed FLOPs)
        fl, by = sum(e[2] for e in ev), sum(e[3] for e in ev)
        ach = fl / (ms * 1e-3) / 1e12
        lib = _lib.load()
        tuning = [int(lib.lpi_get_tuning(k)) for k in range(8)]
        KDESC = {"k256": "gemm256p_kernel (the persistent 256x256 GEMM; the vision and the text tower's GEMM of the same layer op go out as ONE grouped launch, a short "
                         "last round runs as 256x128 half tiles; gemm256_kernel / gemm256_tail_kernel: its one-tile forms)",
                 "k128": "gemm_nt_kernel (128x128 tiles, 4 waves of 64x64, LDS-DMA double buffer: every launch below the 256x256 kernel's tile-count threshold)",
                 "k256x128": "gemm256x128_kernel (launches with 16..159 256x256 tiles)",
                 "few_rows": "gemm_rows_kernel (M <= 512: 32x32 tiles over the whole K range, one launch per op and tower pair)"}
        traffic = mfma_util = tsrc = None
        wl = {"model": a.model, "batch": a.batch, "depth": a.depth}
        import glob
        for pmc in sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc*.json")), reverse=True):
            # separate rocprofv3 --pmc passes of THIS workload (model / batch / depth), dtype, tuning and library build, summarised by tools/pmc_summary.py;
            # a summary of another workload (or of an older build) is never attached: null instead
            pj = json.load(open(pmc))
            if not isinstance(pj, dict):
                continue
            if (pj.get("workload") == wl and pj.get("dtype") == self.dtype and pj.get("tuning") == tuning and pj.get("lib_version") == int(lib.lpi_version())):
                names = {"k256": ("gemm256_kernel", "gemm256_tail_kernel", "gemm256p_kernel"), "k128": ("gemm_nt_kernel",), "k256x128": ("gemm256x128_kernel",),
                         "few_rows": ("gemm_rows_kernel",)}[dom]
                ks = [v for n, v in pj["kernels"].items() if n in names and "hbm_mb_per_launch" in v]
                if ks:      # launch-weighted over the dominant kernel's entry points
                    w = sum(v["launches"] for v in ks)
                    traffic = round(sum(v["hbm_mb_per_launch"] * v["launches"] for v in ks) / w * 1e6)
                    if all("mfma_busy_frac" in v for v in ks):
                        mfma_util = round(sum(v["mfma_busy_frac"] * v["launches"] for v in ks) / w, 4)
                    tsrc = (f"profiles/{os.path.basename(pmc)} (rocprofv3 --pmc, separate passes; FETCH_SIZE/WRITE_SIZE corrected per the guide; "
                            "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024))")
                    break

        def side(b):
            es = groups[b]
            m = t_ms(es)
            return {"launches_per_step": len(es) // nprof, "ms_per_step": round(m / nprof, 3),
                    "achieved_tflops": round(sum(e[2] for e in es) / (m * 1e-3) / 1e12, 2) if m else None, "kernel": KDESC[b]}
        out = {"bound": "mfma", "kernel": KDESC[dom], "dominant_bucket": dom,
               "achieved": round(ach, 2), "peak": PEAK_TF[self.dtype], "unit": "TFLOP/s", "frac": round(ach / PEAK_TF[self.dtype], 4),
               "achieved_fastest_of_n": round(fl / (ms_best * 1e-3) / 1e12, 2),
               "traffic": traffic, "traffic_unit": "HBM bytes per launch", "mfma_util": mfma_util, "traffic_source": tsrc,
               "algorithmic_bytes_per_launch": round(by / len(ev)),
               "launches_per_step": len(ev) // nprof, "avg_launch_us": round(1e3 * ms / len(ev), 2),
               "gemm_ms_per_step": round(ms / nprof, 3), "gemm_gflop_per_step": round(fl / nprof / 1e9, 1),
               "other_gemm_kernels": {b: side(b) for b in groups if b != dom and groups[b]},
               "tuning": tuning,
               "measured": f"HIP events around every GEMM launch of {nrec} extra steps (each launch: the MEDIAN of its {nrec} measurements), towers on one stream "
                           "(kernel alone on the GPU); kernel attribution from lpi_gemm_last_kernel"}
        return out


def eval_path(a, dev, rank, sync, tasks=3, centres=5, n_img=5000, n_txt=25000):
    """The evaluation path of the reference (methods/sprompt.py:433-548, _evaluate_retrieval) on the HIP engine, timed next to the train step:
    per image batch the un-prompted task-id pass (extract_vector), the L1 distance to the tasks' KMeans centres (lpi_l1_task_id) and the
    prompted forward with PER-SAMPLE prompt stacks (visual_interface); the same for captions; then the N_img x N_txt score matrix (f32
    GEMM) and the ranks of the ground truth (lpi_retrieval_rank) at COCO 5k-test size.  Synthetic data and keys; forward only, bf16."""
    import numpy as np
    import torch
    from lpi_amd import _lib
    from lpi_amd.engine import score_matrix
    from lpi_amd.functional import DecomposedPromptFn
    import copy
    a = copy.copy(a)
    a.no_text_shared = True      # inference: un-prompted rows and per-sample prompt stacks — no positions in common, the plain packed layout
    wl = Workload(a, dev, rank, "bf16", True, None)
    enc, B, E = wl.enc, a.batch, wl.cfg.embed_dim
    names = ("dim_1_share", "dim_2_visual", "dim_2_textual", "dim_3_visual", "dim_3_textual")
    with torch.no_grad():
        stacks = [DecomposedPromptFn.apply(*[wl.fac[k] * (1.0 + 0.1 * t) for k in names]) for t in range(tasks)]
        vis_stack = torch.stack([v for v, _ in stacks])          # [tasks, Lyr, P, d_v]
        txt_stack = torch.stack([t for _, t in stacks])
        g = torch.Generator(device="cpu").manual_seed(1234)
        keys_v = torch.nn.functional.normalize(torch.randn(tasks, centres, E, generator=g), dim=-1).to(dev).contiguous()
        keys_t = torch.nn.functional.normalize(torch.randn(tasks, centres, E, generator=g), dim=-1).to(dev).contiguous()
    sel = torch.empty(B, dtype=torch.int32, device=dev)
    s = lambda: torch.cuda.current_stream().cuda_stream  # noqa: E731

    def images_batch():
        with torch.no_grad():
            f = enc.encode_image(wl.images, None)                                                   # get_visual_task_id: extract_vector ...
            _lib.call("lpi_l1_task_id", B, E, tasks, centres, f, E, keys_v, sel, None, s())        # ... nearest task centre (sprompt.py:343-350)
            return enc.encode_image(wl.images, vis_stack[sel.long()], a.depth)                      # visual_interface (slinet.py:215)

    def captions_batch():
        with torch.no_grad():
            f = enc.encode_text(wl.ids, None)
            _lib.call("lpi_l1_task_id", B, E, tasks, centres, f, E, keys_t, sel, None, s())
            return enc.encode_text(wl.ids, txt_stack[sel.long()], a.depth)

    def timed(fn, n, warm):
        for _ in range(warm):
            fn()
        sync()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        sync()
        return (time.perf_counter() - t0) / n

    t_img = timed(images_batch, 10, 2)
    t_txt = timed(captions_batch, 10, 2)
    fi = torch.nn.functional.normalize(torch.randn(n_img, E, generator=g), dim=-1).to(dev)
    ft = torch.nn.functional.normalize(torch.randn(n_txt, E, generator=g), dim=-1).to(dev)
    gt_i = (torch.arange(n_img, dtype=torch.int32).view(-1, 1) * 5 + torch.arange(5, dtype=torch.int32).view(1, -1)).to(dev).contiguous()   # 5 captions per image
    gt_t = (torch.arange(n_txt, dtype=torch.int32) // 5).view(-1, 1).to(dev).contiguous()
    r_i = torch.zeros(n_img, dtype=torch.int32, device=dev)
    r_t = torch.zeros(n_txt, dtype=torch.int32, device=dev)

    def score_and_rank():
        i2t, t2i = score_matrix(fi, ft)
        _lib.call("lpi_retrieval_rank", n_img, n_txt, i2t, n_txt, gt_i, 5, r_i, s())
        _lib.call("lpi_retrieval_rank", n_txt, n_img, t2i, n_img, gt_t, 1, r_t, s())

    t_sc = timed(score_and_rank, 3, 1)
    del wl
    torch.cuda.empty_cache()
    return {"dtype": "bf16", "images_per_s": round(B / t_img, 1), "captions_per_s": round(B / t_txt, 1), "ms_per_image_batch": round(1e3 * t_img, 3),
            "ms_per_caption_batch": round(1e3 * t_txt, 3), "batch": B, "tasks": tasks,
            "score_and_rank_ms": round(1e3 * t_sc, 3), "score_shape": [n_img, n_txt],
            "workload": "methods/sprompt.py:433-548 per batch: un-prompted task-id pass + L1 distance to tasks x 5 KMeans centres + prompted forward with "
                        "per-sample prompt stacks (depth as the train step); then the f32 score matrix and ground-truth ranks at COCO 5k-test size"}


def attention_gflop(cfg, B, P, text_rows, fwd_only, text_shared=0):
    """Matrix FLOPs the attention kernels execute per step (2 x MACs, dense L x L per head over the rows actually computed; the causal text
    tower's masked tiles count half): forward 4 L^2 d per sample and layer, backward 8 L^2 d (the recomputed scores are not counted: the
    model-FLOP convention of SURVEY 8(d)); the last block attends from the pooled row only (attn_pooled.hip), the first block's backward covers
    the prompt rows only."""
    Lv, dv, nv = 1 + P + cfg.n_patches, cfg.vision_width, cfg.vision_layers
    Lt, dt, nt = text_rows, cfg.transformer_width, cfg.transformer_layers
    fv = 4.0 * Lv * Lv * dv * B * (nv - 1)
    ft = 0.5 * 4.0 * Lt * Lt * dt * B * (nt - 1)
    if text_shared:      # shared prefix: a sample's Lt own rows attend to the shared keys (all of them) and causally to each other; the shared sequence once
        ft = 4.0 * (Lt * (text_shared + 0.5 * Lt) * B + 0.5 * text_shared * text_shared) * dt * (nt - 1)
    if fwd_only:
        return (fv + ft) / 1e9
    return (fv + ft + 2.0 * (fv + ft) * (max(nv - 2, 0) / max(nv - 1, 1))) / 1e9


def run_record(a, dev, proc_rank, sync, dtype, fwd_only, steps, warm, roofline=True, **over):
    """One more workload of the same bench (a precision mode, the forward-only configuration, another model) as a first-class record: the same
    timing contract (wall clock around exactly `steps` steps, median of per-step HIP events) and, for training steps, its own roofline block from
    its own instrumented pass against the matching peak."""
    import copy
    import numpy as np
    import torch
    import gc
    a2 = copy.copy(a)
    for k, v in over.items():
        setattr(a2, k, v)
    # the previous record's engine (weights, workspace: GBs) is garbage by now: collect it and hand its memory back BEFORE the timed steps — left to the
    # allocator and the cyclic collector, the frees (each a device synchronisation) landed inside one timed step of the next record (round 4: the forward-only
    # record's mean step read 13.85 ms beside a median of 10.36)
    gc.collect()
    torch.cuda.empty_cache()
    wl = Workload(a2, dev, proc_rank, dtype, fwd_only, None)
    el, per = wl.run(steps, warm, sync)
    B = a2.batch
    v = B * steps / el
    gfs = GFLOP_PER_PAIR.get(a2.model)
    rec = {"dtype": dtype, "value": round(v, 2), "unit": "pairs/s", "steps": steps, "warmup": warm, "ms_per_step": round(1e3 * el / steps, 3),
           "median_ms_per_step": round(float(np.median(per)), 3),
           "step_mfma_frac": None if gfs is None else round(v * gfs[1 if fwd_only else 0] * 1e9 / (PEAK_TF[dtype] * 1e12), 4),
           "peak_tflops": PEAK_TF[dtype]}
    if roofline and not fwd_only:
        rec["roofline"] = wl.gemm_roofline()
        hw = wl.gemm_gflop_all + attention_gflop(wl.cfg, B, 16, wl.text_rows, fwd_only, wl.text_shared)
        rec["hw_flop_frac"] = round(hw * 1e9 / (1e-3 * float(np.median(per))) / (PEAK_TF[dtype] * 1e12), 4)
        rec["executed_gflop_per_step"] = round(hw, 1)
    del wl
    torch.cuda.empty_cache()
    return rec


def parity_block(a, dev):
    """Parity on the same line as the headline.  No oracle here: the reference's own outputs come from the committed fixture."""
    import gc
    import numpy as np
    import torch
    from lpi_amd import synth
    from lpi_amd.engine import DualEncoder
    from lpi_amd.step import train_step
    cfg = synth.CONFIGS["ViT-B/16"]
    gpath = os.path.join(REPO, "tests", "golden", "vitb16_d3_patched.npz")
    out = {}

    def run(enc, img, ids, depth):
        fac = {k: torch.from_numpy(v).to(dev).requires_grad_(True) for k, v in synth.prompt_factors(9, 16, cfg.vision_width, cfg.transformer_width).items()}
        o = train_step(enc, img, ids, fac, depth)
        torch.cuda.synchronize()
        return o, fac
    mx = lambda x, y: float(np.abs(np.asarray(x, dtype=np.float64) - np.asarray(y, dtype=np.float64)).max())  # noqa: E731
    sd = synth.clip_state_dict(cfg)
    enc32 = DualEncoder(cfg, sd, dtype="f32", device=dev)
    if os.path.isfile(gpath):
        g = dict(np.load(gpath, allow_pickle=False))
        img = torch.from_numpy(synth.images(8, cfg.image_resolution)).to(dev)
        from lpi_amd.engine import PackedIds as _PK
        # the reference's own text layout (77 columns), then the HEADLINE's (packed at every caption's EOT, the 17 common positions stored once): the layout is
        # exact, so the f32 step meets the same bar on it
        for key, text in (("f32_vs_reference_fixture", torch.from_numpy(g["token_ids"]).to(dev)),
                          ("f32_on_the_headline_text_layout_vs_reference_fixture", _PK(g["token_ids"], 0 if a.no_text_shared else 17).to(dev))):
            o, fac = run(enc32, img, text, 3)
            lg = (enc32.logit_scale_exp * o["img_f"] @ o["txt_f"].t()).cpu().numpy()
            out[key] = {
                "fixture": "tests/golden/vitb16_d3_patched.npz (imported reference, deep-prompt guard patched: SURVEY F1; ViT-B/16, 8 pairs, depth 3, r 4)",
                "max_abs_logit_err": mx(lg, g["logits"]), "max_abs_feature_err": max(mx(o["img_f"].cpu().numpy(), g["img_f"]), mx(o["txt_f"].cpu().numpy(), g["txt_f"])),
                "base_loss_err": abs(float(o["base_loss"]) - float(g["base_loss"])),
                "max_rel_factor_grad_err": max(mx(fac[k].grad.cpu().numpy(), g["grad." + k]) / float(np.abs(g["grad." + k]).max()) for k in synth.PROMPT_NAMES),
                "bar": "1e-4 (logits, losses), 1e-3 relative (factor gradients): tests/test_model_gpu.py, tests/test_shared_prefix_gpu.py"}
    B = a.batch
    img = torch.from_numpy(synth.images(B, cfg.image_resolution)).to(dev)
    ids = torch.from_numpy(synth.token_ids(B)).to(dev)
    # round 6: the benchmarked batch itself went through the imported reference once (tests/golden/gen_golden.py --only bs256): when this run is that
    # configuration, every mode below is also measured against the REFERENCE's outputs at 256 pairs, not only against the f32 HIP step
    big = None
    bpath = os.path.join(REPO, "tests", "golden", {1: "vitb16_bs256_d1.npz", 3: "vitb16_bs256_d3_patched.npz"}.get(a.depth, "-"))
    if B == 256 and a.rank == 4 and a.prompt_layers == 9 and os.path.isfile(bpath):
        big = dict(np.load(bpath, allow_pickle=False))

    def vs_reference(enc, o, fac, layout_name):
        i_f, t_f = o["img_f"].double().cpu().numpy(), o["txt_f"].double().cpu().numpy()
        lg = float(enc.logit_scale_exp) * i_f @ t_f.T
        err = mx(lg, big["logits"])
        gr = {k: fac[k].grad.double().cpu().numpy() for k in synth.PROMPT_NAMES}
        top1 = float(np.mean(np.concatenate([(lg.argmax(1) == big["top5_i2t"][:, 0]), (lg.T.argmax(1) == big["top5_t2i"][:, 0])])))
        return {"fixture": os.path.relpath(bpath, REPO) + " (imported reference, f32 CPU, this batch; depth > 1: deep-prompt guard patched, SURVEY F1)",
                "text_layout": layout_name, "max_abs_feature_err": max(mx(i_f, big["img_f"]), mx(t_f, big["txt_f"])), "max_abs_logit_err": err,
                "base_loss_err": abs(float(o["base_loss"]) - float(big["base_loss"])), "alignment_loss_err": abs(float(o["alignment_loss"]) - float(big["alignment_loss"])),
                "max_rel_factor_grad_err": max(mx(gr[k], big["grad." + k]) / float(np.abs(big["grad." + k]).max()) for k in synth.PROMPT_NAMES),
                "min_factor_grad_cosine": min(float((gr[k] * big["grad." + k]).sum() / (np.linalg.norm(gr[k]) * np.linalg.norm(big["grad." + k]))) for k in synth.PROMPT_NAMES),
                "top1_agreement_with_reference": top1, "bar": "tests/test_reference_bs256_gpu.py"}
    o32, f32 = run(enc32, img, ids, a.depth)
    if big is not None:
        out[f"f32_vs_reference_fixture_bs{B}"] = vs_reference(enc32, o32, f32, "77 columns")
    o32 = {k: v.clone() for k, v in o32.items()}
    g32 = {k: f32[k].grad.double().cpu() for k in synth.PROMPT_NAMES}
    del enc32
    gc.collect()
    torch.cuda.empty_cache()
    # the throughput modes run in the text layout of the headline (packed at every caption's EOT, the 17 common positions stored once) — against the f32
    # step on the reference's own layout (all 77 columns) above
    from lpi_amd.engine import PackedIds
    ids_host = synth.token_ids(B)
    layout = ids if (a.no_text_trim or a.no_text_pack) else PackedIds(ids_host, 0 if a.no_text_shared else 17).to(dev)
    for mode in ("bf16", "f16"):
        enc = DualEncoder(cfg, sd, dtype=mode, device=dev)
        ob, fb = run(enc, img, layout, a.depth)
        rel = max(float((fb[k].grad.double().cpu() - g32[k]).abs().max() / g32[k].abs().max()) for k in synth.PROMPT_NAMES)
        cos = min(float((fb[k].grad.double().cpu() * g32[k]).sum() / (fb[k].grad.double().cpu().norm() * g32[k].norm())) for k in synth.PROMPT_NAMES)
        l32 = enc.logit_scale_exp * o32["img_f"] @ o32["txt_f"].t()
        lb = enc.logit_scale_exp * ob["img_f"] @ ob["txt_f"].t()
        out[f"{mode}_vs_f32_hip_bs{B}"] = {"max_abs_feature_err": max(float((ob["img_f"] - o32["img_f"]).abs().max()), float((ob["txt_f"] - o32["txt_f"]).abs().max())),
                                           "max_abs_logit_err": float((lb - l32).abs().max()), "base_loss_rel_err": abs(float(ob["base_loss"]) - float(o32["base_loss"])) / abs(float(o32["base_loss"])),
                                           "min_factor_grad_cosine": cos, "max_rel_factor_grad_err": rel, "top1_agreement": float((lb.argmax(1) == l32.argmax(1)).float().mean()),
                                           "text_layout": "77 columns" if layout is ids else ("packed" + (", shared prefix 17" if layout.shared else "")) + " (f32 leg: 77 columns)",
                                           "bar": "features 5e-3 (bf16) / 1.5e-3 (f16), gradient cosine 0.9995: tests/test_fullsize_gpu.py"}
        if big is not None:
            out[f"{mode}_vs_reference_fixture_bs{B}"] = vs_reference(enc, ob, fb, "77 columns" if layout is ids else ("packed" + (", shared prefix 17" if layout.shared else "")))
        del enc
        gc.collect()
        torch.cuda.empty_cache()
    return out


def plugin_step(a, dev, sync, steps=60, warm=10, **over):
    """The plugin's hot loop as the reference's caller runs it (methods/sprompt.py:290-334: DataLoader batch -> images.cuda() -> SliNet.forward on caption
    strings -> cal_loss -> backward -> optimizer.step), timed like every other record: wall clock around exactly `steps` iterations of SPrompts.train_epoch,
    synchronised on both sides.  The dataset yields host f32 images (views of a pool of 512 distinct N(0,1) images: generating 150 k normals per item
    would time numpy) and caption strings; tokenisation uses CLIP's merge table if the box has one, else a synthetic table of the same format."""
    import gc
    import numpy as np
    import torch
    from torch.utils.data import DataLoader
    from lpi_amd import _lib
    from lpi_amd.retrieval.methods.sprompt import SPrompts
    from lpi_amd.retrieval.utils.data import SyntheticCoco, collate_keep_images
    from lpi_amd.synth_bpe import ensure_vocab
    vocab = ensure_vocab()
    gc.collect()
    torch.cuda.empty_cache()
    B = a.batch
    args = json.load(open(os.path.join(REPO, "lpi_amd", "retrieval", "configs", "lpi", "coco_lpi.json")))
    args.update(device=[dev], compute_dtype=a.dtype, honor_prompt_depth=True, prompt_depth=a.depth, batch_size=B, epochs=1, num_workers=0, r=a.rank,
                backbonename=a.model, pipeline_timing=True)
    args.update(over)
    m = SPrompts(args)
    net = m._network
    net.update_fc(0)                                            # task 0 (numtask = 1), as incremental_train does
    pf = args.get("pixel_format", "f32")
    ds = SyntheticCoco((steps + warm + 8) * B, [0], net.clip_cfg.image_resolution, seed=0, captions="strings", image_pool=512, pixel_format=pf)
    loader = DataLoader(ds, batch_size=B, shuffle=False, num_workers=0, collate_fn=collate_keep_images)
    optimizer, _ = m._setup_training()
    t, host, h2d, rows, evs = {}, [], [], [], []
    # the MAIN thread's side of an iteration: how long it takes to enqueue a step (Python generators + ctypes: no device wait in it) and how long it sits
    # between two steps (hand-over of the next batch, the loss log) — against the device's step time this says how far ahead of the GPU the host runs
    enq, stamps = [], []
    if args.get("fused_step", True):
        inner = net.train_step

        def timed_step(*aa, **kw):
            t0 = time.perf_counter()
            r = inner(*aa, **kw)
            enq.append((t0, time.perf_counter()))
            return r
        net.train_step = timed_step

    def on_step(i, batch, out):
        stamps.append(time.perf_counter())
        if i >= warm - 1:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            evs.append(e)
        if i == warm - 1:
            sync()
            t["n0"] = _lib.launch_count()
            t["t0"] = time.perf_counter()
        elif i >= warm:
            if hasattr(batch, "host_ms"):
                host.append(batch.host_ms)
                h2d.append(batch.h2d)
                rows.append(getattr(batch.text, "rows", 0))
            if i == warm + steps - 1:
                t["n1"] = _lib.launch_count()
                sync()
                t["t1"] = time.perf_counter()
                return True
        return False

    m.train_epoch(loader, optimizer, 0, None, on_step)
    el = t["t1"] - t["t0"]
    rec = {"dtype": a.dtype, "value": round(B * steps / el, 2), "unit": "pairs/s", "steps": steps, "warmup": warm, "ms_per_step": round(1e3 * el / steps, 3),
           "launches_per_step": (t["n1"] - t["n0"]) // steps,
           "loop": "SPrompts.train_epoch: " + ("BatchPipeline (pinned staging, side-stream H2D, tokenise ahead) + fused SliNet.train_step + FlatSGD"
                                               if args.get("prefetch", True) else "images.to(device) + net(images, captions) -> cal_loss -> backward + FlatSGD (reference order)"),
           "input": f"DataLoader(SyntheticCoco: host {pf} images [3,{net.clip_cfg.image_resolution},{net.clip_cfg.image_resolution}] from a pool of 512, caption strings), bs={B}, "
                    "num_workers=0", "bpe_table": "synthetic" if "lpi_synthetic_bpe" in vocab else "clip"}
    # step-to-step time on the device inside the loop (HIP events recorded behind every iteration): what the loop costs the GPU, next to the wall clock
    per = [evs[j].elapsed_time(evs[j + 1]) for j in range(len(evs) - 1)]
    rec["median_ms_per_step"] = round(float(np.median(per)), 3)
    rec["ms_per_step_p10_p90"] = [round(float(np.percentile(per, 10)), 3), round(float(np.percentile(per, 90)), 3)]
    if len(enq) > warm + 2:
        d = np.array([1e3 * (b - a_) for a_, b in enq[warm:]])
        gap = np.array([1e3 * (enq[j + 1][0] - enq[j][1]) for j in range(warm, len(enq) - 1)])
        rec["main_thread_ms_per_iteration"] = {"enqueue_step_median": round(float(np.median(d)), 3), "enqueue_step_p90": round(float(np.percentile(d, 90)), 3),
                                               "between_steps_median": round(float(np.median(gap)), 3), "between_steps_p90": round(float(np.percentile(gap, 90)), 3)}
    if host:
        rec["producer_ms_per_batch"] = {k: round(float(np.mean([h[k] for h in host])), 3) for k in host[0]}
        rec["producer_ms_per_batch"]["total_without_waits"] = round(sum(v for k, v in rec["producer_ms_per_batch"].items() if k not in ("slot_wait",)), 3)
        rec["h2d_ms_per_batch"] = round(float(np.mean([e0.elapsed_time(e1) for e0, e1 in h2d if e0 is not None])), 3)
        rec["h2d_bytes_per_batch"] = B * 3 * net.clip_cfg.image_resolution ** 2 * (1 if pf == "u8" else 4)
        rec["text_rows_computed"] = round(float(np.mean(rows)) / B, 2)
    # the split of one step on the device (HIP events at the phase boundaries of a few more steps, on the loop's own batches)
    marks_all = []
    it = iter(DataLoader(ds, batch_size=B, shuffle=False, num_workers=0))
    staged = []
    for _ in range(7):                                           # resident first, so that the instrumented steps run back to back (no host copy between them)
        img, caps = next(it)[:2]
        staged.append((img.to(dev), net.prepare_text(list(caps)).to(dev)))
    torch.cuda.synchronize()
    for img, ids in staged:
        marks = []
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append(("start", e))
        net.train_step(img, ids, flat_grad=optimizer.flat_grad, grad_views=optimizer.grad_views, marks=marks)
        optimizer.step()
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append(("optimiser", e))
        marks_all.append(marks)
    marks_all = marks_all[2:]                                    # the first two start on an idle GPU (launch latency exposed)
    torch.cuda.synchronize()
    rec["device_ms"] = {name: round(float(np.median([mk[j][1].elapsed_time(mk[j + 1][1]) for mk in marks_all])), 3)
                        for j, name in enumerate(n for n, _ in marks_all[0][1:])}
    del m, net, optimizer, loader, ds
    gc.collect()
    torch.cuda.empty_cache()
    return rec


def packed_ids_host_cost(B, dev, n=20):
    """Per-batch host cost of the packed text layout (engine.PackedIds: argmax / cumsum over the tokenizer's [B, 77] ids and the upload of the index
    arrays) — host work next to tokenisation, outside the timed step like it, reported so that it is not hidden."""
    import torch
    from lpi_amd import synth
    from lpi_amd.engine import PackedIds
    ids_host = synth.token_ids(B, seed=synth.TOKEN_SEED)
    PackedIds(ids_host).to(dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        PackedIds(ids_host).to(dev)
    torch.cuda.synchronize()
    return round(1e6 * (time.perf_counter() - t0) / n, 1)


def main():
    a = parse_args()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(self_launch(a))

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    cpus = pin_rank_to_cpus(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)), a.share_gpu, not a.no_affinity)      # before the first GPU call of this process
    if cpus is not None:
        torch.set_num_threads(max(1, min(len(cpus), 16)))
    dev_index = 0 if a.share_gpu else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    exchange = None
    if world > 1 or a.force_dist:       # --force-dist: exercise the RCCL path on a 1-GPU box
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if a.share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        from lpi_amd.dp import Exchange
        exchange = Exchange(timing=True, local_loss=a.local_loss, gather_with_grad=a.gather_with_grad)
    def sync():
        if exchange is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def rank_max(x):
        if exchange is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev if exchange.device_collectives else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    wl = Workload(a, dev, rank, a.dtype, a.fwd_only, exchange)
    cfg, B = wl.cfg, a.batch
    el, per = wl.run(a.steps, a.warmup, sync)
    el = rank_max(el)
    pairs_s = world * B * a.steps / el
    median_ms = rank_max(float(np.median(per)))
    collectives = None
    if exchange is not None:
        torch.cuda.synchronize()
        tt = exchange.timing or []
        per_step = max(1, len(tt) // max(1, a.steps + a.warmup))      # collectives per step: 2 (3 with gradients through the gathered features; 1 forward only)
        tl = tt[-per_step * a.steps:]
        # device collectives (RCCL) are timed with HIP events; host-staged ones (gloo, ranks sharing a GPU) carry no per-collective time
        collectives = {k: round(1e3 * float(np.mean([e0.elapsed_time(e1) for kk, e0, e1 in tl if kk == k])), 1)
                       for k in sorted({kk for kk, _, _ in tl})} or None
        E2 = 2 * cfg.embed_dim
        nfac = sum(int(v.numel()) for v in wl.fac.values())
        mine = {"rank": rank, "mean_us_per_step": collectives, "median_ms_per_step": round(float(np.median(per)), 3), "cpus": None if cpus is None else len(cpus)}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)                     # after the timed region: a few hundred bytes of bookkeeping per rank
        meds = [r["median_ms_per_step"] for r in per_rank if r is not None]
        collectives = {"mean_us_per_step": collectives, "per_rank": per_rank, "backend": dist.get_backend(), "observed_world_size": dist.get_world_size(),
                       # round 6, first-contact diagnostics: the spread of the ranks' own median step (a straggler GPU or a starved host shows here) ...
                       "per_rank_median_ms": {"min": min(meds), "max": max(meds), "spread_pct": round(100.0 * (max(meds) - min(meds)) / min(meds), 2)},
                       "gpu_numa_nodes": gpu_numa_nodes(),
                       "messages": f"one all_gather_into_tensor of img_f||txt_f [B, {E2}] f32 ({B * E2 * 4} bytes per rank) + one all_reduce(SUM) of the "
                                   f"{nfac} factor gradients"
                                   + (f" + one reduce_scatter_tensor of the key gradients [W B, {E2}] f32" if a.gather_with_grad else "")}
        exchange.timing = None
        # the same step WITHOUT the exchange on every rank at once (each rank its own contrastive matrix, no collective): what the collectives and the
        # larger loss matrix cost the job — a self-check beside the driver's own N = 1 run, not a scaling claim
        if not a.fwd_only:
            wl.exchange = None
            el0, _ = wl.run(max(4, a.steps // 2), 2, sync)
            el0 = rank_max(el0)
            v0 = world * B * max(4, a.steps // 2) / el0
            collectives["without_exchange"] = {"value": round(v0, 2), "unit": "pairs/s", "steps": max(4, a.steps // 2),
                                               "ratio_with_exchange": round(pairs_s / v0, 4),
                                               "note": "all ranks stepping at once with exchange = None (local 256 x 256 loss, no all-gather / all-reduce)"}
            # ... and ONE rank stepping alone while the others wait at a barrier: the N = 1 value of THIS run on THIS node.  It should match the driver's
            # single-GPU line within box noise (5 %); `without_exchange` / W below it = what W busy GPUs cost each other (host, power, fabric), and
            # `value` below `without_exchange` = what the exchange and the W x larger loss matrix cost.
            n_solo = max(4, a.steps // 2)
            el1 = None
            if rank == 0:
                el1, per1 = wl.run(n_solo, 2, lambda: torch.cuda.synchronize())
            dist.barrier()
            wl.exchange = exchange
            if rank == 0:
                v1 = B * n_solo / el1
                collectives["single_rank_alone"] = {"value": round(v1, 2), "unit": "pairs/s", "steps": n_solo, "median_ms_per_step": round(float(np.median(per1)), 3),
                                                    "all_ranks_without_exchange_per_gpu_ratio": round(v0 / world / v1, 4),
                                                    "job_per_gpu_ratio": round(pairs_s / world / v1, 4),
                                                    "note": "rank 0 stepping alone (exchange = None), the other ranks idle at a barrier: compare with the N = 1 line"}

    roofline = None if a.no_roofline else wl.gemm_roofline()
    from lpi_amd import _lib as _L
    torch.cuda.synchronize()
    n0 = _L.launch_count()
    wl.step()
    torch.cuda.synchronize()
    launches_per_step = _L.launch_count() - n0      # library launches of one steady-state step (no other kernel runs in it: tests/test_round4_gpu.py)

    hw_gflop = None
    if roofline is not None and not a.fwd_only:
        hw_gflop = wl.gemm_gflop_all + attention_gflop(cfg, B, 16, wl.text_rows, False, wl.text_shared)
    extras = {}
    if world == 1 and not a.no_extras and not a.fwd_only and a.dtype == "bf16":
        del wl.opt
        # the same workload in the parity mode (f32 operands: meets the 1e-4 bar, roofline vs the 157.3 TF f32 MFMA peak) ...
        extras["parity_mode"] = run_record(a, dev, rank, sync, "f32", False, 20, 3)
        extras["parity_mode"]["note"] = "f32-in / f32-accumulate MFMA; the mode whose logits / grads meet the 1e-4 parity bar"
        # ... and BASELINE.json configs[1]: forward-only encoders + cosine matrix, bf16
        extras["fwd_only"] = run_record(a, dev, rank, sync, "bf16", True, 20, 3)
        extras["fwd_only"]["workload"] = "BASELINE.json configs[1]: ViT-B/16 bs=256 prompt_depth=3 r=4, fwd-only encoder + cosine-sim matrix"
        # ... and the f16 operand mode: fp16 MFMA operands / activations in the forward (the reference's own arithmetic type), bf16 backward
        extras["f16_mode"] = run_record(a, dev, rank, sync, "f16", False, 20, 3)
        extras["f16_mode"]["note"] = ("compute_dtype='f16': v_mfma_f32_16x16x32_f16 forward, bf16 gradient stream; on the ViT-B/16 fixture 4.6x lower "
                                      "feature error and 4x lower logit error than the bf16 line (tests/test_model_gpu.py); the fp16 MFMA runs "
                                      "5-9 % slower than the bf16 one on the same GEMM shapes (power-limited clock), hence not the default")
        # ... and the headline workload with the 17 common text positions stored per sample (the layout of rounds 3-4), for the A/B on this box
        if not a.no_text_shared:
            extras["without_shared_text_prefix"] = run_record(a, dev, rank, sync, "bf16", False, 20, 3, roofline=False, no_text_shared=True)
            extras["without_shared_text_prefix"]["note"] = "engine.PackedIds without shared=17 (bench.py --no-text-shared): every caption carries its own SOT + 16 context rows"
        # ... and BASELINE.json configs[4]'s per-GPU workload: ViT-L/14, 512 pairs, prompt_depth 12, r 8, bf16
        if a.model == "ViT-B/16":
            extras["vit_l14"] = run_record(a, dev, rank, sync, "bf16", False, 8, 2, model="ViT-L/14", batch=512, depth=12, rank=8, prompt_layers=12)
            extras["vit_l14"]["workload"] = "BASELINE.json configs[4] on one GPU: ViT-L/14 dual encoder bs=512/GPU prompt_depth=12 r=8, fwd+bwd + SGD step"
        # ... and the step inside the reference's caller loop: DataLoader -> H2D -> tokenise -> SliNet -> FlatSGD (methods/sprompt.py:290-334)
        import contextlib
        with contextlib.redirect_stdout(sys.stderr):       # the learner prints its trainable set: stdout carries the ONE JSON line only
            extras["plugin_step"] = plugin_step(a, dev, sync)
            extras["plugin_step"]["vs_bare_step"] = round(extras["plugin_step"]["value"] / pairs_s, 4)
            extras["plugin_step_reference_order"] = plugin_step(a, dev, sync, steps=15, warm=4, prefetch=False, fused_step=False)
            # ... and with the dataset handing over decoded uint8 pixels (ToTensor + Normalize inside lpi_patchify_u8, bit for bit: a quarter of the H2D bytes)
            extras["plugin_step_u8"] = plugin_step(a, dev, sync, steps=40, warm=8, pixel_format="u8")
            extras["plugin_step_u8"]["vs_bare_step"] = round(extras["plugin_step_u8"]["value"] / pairs_s, 4)
        if a.model == "ViT-B/16":
            extras["parity"] = parity_block(a, dev)
        extras["eval_path"] = eval_path(a, dev, rank, sync)
        extras["packed_ids_host_us_per_batch"] = packed_ids_host_cost(B, dev)

    if rank == 0:
        gfs = GFLOP_PER_PAIR.get(a.model)
        gf = None if gfs is None else gfs[1 if a.fwd_only else 0]
        out = {
            "metric": (f"image-text pairs/sec fwd+bwd ({a.model}, bs{B}/GPU)" if not a.fwd_only
                       else f"image-text pairs/sec fwd-only encoder + cosine matrix ({a.model}, bs{B}/GPU)"),
            "value": round(pairs_s, 2), "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * el / a.steps, 3), "median_ms_per_step": round(median_ms, 3),
            "value_at_median": round(world * B / (median_ms * 1e-3), 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": ("BASELINE.json configs[2]: " if (a.model == "ViT-B/16" and B == 256 and a.depth == 3 and a.rank == 4 and not a.fwd_only) else "")
                       + f"{a.model} dual encoder bs={B}/GPU prompt_depth={a.depth} r={a.rank} P=16, "
                       + ("fwd-only + cosine matrix" if a.fwd_only else "fwd+bwd incl. DecomposedPrompt grads + SGD step"),
                       "global_batch": world * B, "image": f"{cfg.image_resolution}x{cfg.image_resolution}", "tokens": cfg.context_length,
                       # rows per caption actually computed: the rows behind a caption's EOT are dead under the causal mask and skipped, exactly
                       # (packed: every caption cut at its OWN EOT; --no-text-pack: at the longest one; --no-text-trim: all 77)
                       "text_rows_computed": round(wl.text_rows, 2),
                       "text_layout": "77 columns" if a.no_text_trim else ("cut at the longest caption" if a.no_text_pack else ("packed (engine.PackedIds)" + (", SOT + 16 broadcast context slots stored once for the batch (shared=17)" if wl.text_shared else ""))),
                       "towers": "one after the other" if (a.no_lockstep or a.overlap) else "lock step, GEMMs of one layer op grouped in one launch",
                       "parallelism": f"dp{world}" + (" (ranks share one GPU, gloo, host-staged messages)" if a.share_gpu and world > 1 else ""),
                       "dp_mode": None if world == 1 and exchange is None else
                       f"gather_features(local_loss={a.local_loss}, gather_with_grad={a.gather_with_grad})",
                       "weights": "synthetic (numpy Philox, CLIP-init scales), frozen",
                       "precision": "bf16 MFMA operands, f32 accumulate, fp16 residual stream, bf16 gradient stream; parity at 1e-4 is a property of "
                                    "the f32 mode (parity_mode), not of this line" if a.dtype == "bf16" else "f32 MFMA operands and accumulate (parity mode)"},
            # model-FLOP utilisation (MFU): SURVEY 8(d)'s algorithmic 89.68 GFLOP per pair (77 text rows, whole first and last blocks) / time / peak
            "step_mfma_frac": None if gf is None else round(pairs_s / world * gf * 1e9 / (PEAK_TF[a.dtype] * 1e12), 4),
            "step_mfma_frac_kind": "MFU: the reference model's algorithmic FLOPs (SURVEY 8(d)); hw_flop_frac counts the FLOPs the kernels execute",
            # hardware-FLOP utilisation (HFU): what the GEMM launches of a step execute (packed text rows, pooled last block, prompt-row first-block
            # backward) + the attention kernels' matrix FLOPs, / median step time / peak
            "hw_flop_frac": None if hw_gflop is None else round(hw_gflop * 1e9 / (median_ms * 1e-3) / (PEAK_TF[a.dtype] * 1e12), 4),
            "executed_gflop_per_step": None if hw_gflop is None else round(hw_gflop, 1),
            "roofline": roofline,
            "launches_per_step": launches_per_step,
        }
        out.update(extras)
        if collectives is not None:
            out["collectives"] = collectives
        out["cpu_baseline"] = cpu_baseline(cfg, a.depth, bs256=a.cpu_baseline_bs256) if (world == 1 and not a.no_cpu_baseline) else None
    line = json.dumps(out) if rank == 0 else None
    try:
        if exchange is not None:
            dist.barrier()
            dist.destroy_process_group()
    finally:
        if line is not None:      # after the process group is gone: RCCL's banner / teardown chatter cannot follow the one JSON line
            sys.stdout.flush()
            print(line, flush=True)


if __name__ == "__main__":
    main()

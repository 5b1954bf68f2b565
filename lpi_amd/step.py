"""One training step of the LPI hot path on the HIP engine, written the way the reference's hot loop runs it
(methods/sprompt.py:297-311: forward -> cal_loss -> sum of losses -> backward), minus the optimiser.

Used by bench.py, __graft_entry__.smoke() and the parity tests; the plugin surface (lpi_amd/retrieval) reaches the
same Functions through SliNet.forward / SliNet.cal_loss.
"""
from __future__ import annotations

import torch

from . import engine as E
from .functional import AlignLossFn, ClipLossFn, DecomposedPromptFn, EncodeBothFn, EncodeImageFn, EncodeTextFn, contrastive_loss_and_grads
from .synth import PROMPT_NAMES

_CP_ORDER = ("dim_1_share", "dim_2_visual", "dim_2_textual", "dim_3_visual", "dim_3_textual")


def forward_loss(enc, images, ids, factors: dict, depth: int = 1, gather=None, align_weight: float = 0.1, overlap_towers: bool = False,
                 vision_lanes: int = 1, text_lanes: int = 1, lockstep: bool = True, exchange=None, grad_out=None, losses=True):
    """factors: the five DecomposedPrompt parameters (device tensors, requires_grad as wanted).
    Returns (losses dict of 0-d tensors, img_f, txt_f, vis, txt, logits).
"""
    vis, txt = DecomposedPromptFn.apply(*[factors[k] for k in _CP_ORDER], 1.0, grad_out)
    if overlap_towers:
        # The towers are independent until the similarity matrix, and so are micro-batches of one tower: each (tower, micro-batch)
        # "lane" runs on its own HIP stream with its own workspace.  One lane's HBM-bound kernels (LayerNorm, attention, GEMM
        # epilogues) then overlap another lane's MFMA-bound GEMM main loops, and lanes fill each other's partial last round of
        # GEMM tiles.  autograd replays each node's backward on the stream its forward ran on, so the backward overlaps the same
        # way; the factor gradients are sums over the batch, so splitting it changes nothing but the summation order.
        main = torch.cuda.current_stream()
        B = images.shape[0]
        nv = vision_lanes if B % vision_lanes == 0 else 1
        ntx = text_lanes if B % text_lanes == 0 else 1
        jobs = [("v", k, nv) for k in range(nv)] + [("t", k, ntx) for k in range(ntx)]
        outs = {"v": [], "t": []}
        for li, (kind, k, n) in enumerate(jobs):
            eng = enc.lane(li)
            sl = slice(k * (B // n), (k + 1) * (B // n))
            st = main if li == 0 else eng.stream
            if li:
                st.wait_stream(main)
            with torch.cuda.stream(st):
                f = EncodeImageFn.apply(eng, images[sl], vis, depth) if kind == "v" else EncodeTextFn.apply(eng, ids[sl], txt, depth)
            if li:
                f.record_stream(main)
            outs[kind].append(f)
        for li in range(1, len(jobs)):
            main.wait_stream(enc.lane(li).stream)
        img_f = outs["v"][0] if nv == 1 else torch.cat(outs["v"])
        txt_f = outs["t"][0] if ntx == 1 else torch.cat(outs["t"])
    elif lockstep:
        # one stream, the towers in lock step: their GEMMs of the same layer op go out as ONE grouped persistent launch
        img_f, txt_f = EncodeBothFn.apply(enc, images, ids, vis, txt, depth)
    else:
        img_f = EncodeImageFn.apply(enc, images, vis, depth)
        txt_f = EncodeTextFn.apply(enc, ids, txt, depth)
    if not losses:
        return None, img_f, txt_f, vis, txt
    base = ClipLossFn.apply(img_f, txt_f, enc.logit_scale_exp, gather, exchange)
    losses = {"base_loss": base, "alignment_loss": AlignLossFn.apply(vis, txt, 0.01, align_weight)}
    return losses, img_f, txt_f, vis, txt


def train_step(enc, images, ids, factors: dict, depth: int = 1, exchange=None, align_weight: float = 0.1, overlap_towers: bool = False,
               vision_lanes: int = 1, text_lanes: int = 1, lockstep: bool = True, flat_grad=None, grad_views=None, task_term=None, marks=None):
    """forward + losses + backward; leaves the gradients in factors[k].grad and returns the forward outputs.

    exchange: a ``dp.Exchange`` for data-parallel runs (features all-gathered for the global contrastive matrix, factor
    gradients SUM all-reduced; the data-independent alignment term is scaled by 1/W so that it counts once).
    flat_grad / grad_views (optim.flatten): the factor gradients are written into that flat vector (the .grad tensors are its slices), the
    all-reduce runs on it as it is and optim.FlatSGD steps in one launch.

    The loss kernels produce the loss values AND their gradients w.r.t. features / prompts in one pass, so the towers' backward is seeded with
    those gradients directly (torch.autograd.backward on the features and the prompt stacks): no scalar loss graph, no `grad * g` kernels.  The
    weights of the sum (sprompt.py:308: the plain sum of the dict; data parallel: see above) are host numbers folded into the seeds.

    task_term: callable(vis, txt, dvis_buf, dtxt_buf, weight_scale) -> loss tensor(s) — the task loss of a continual session (slinet.py:160-162, 167-183: only when
    numtask != 1).  It is data-independent like the alignment term: it ADDS its dense gradient onto the seeded prompt-gradient buffers (needs the seeded
    path) and is scaled by 1/W under data parallelism; its (unscaled) value comes back as out["task_loss"].
    marks: a list that receives (name, HIP event) pairs at the phase boundaries (bench.py's split of the plugin step); None = no events.

    Seeded mode hands autograd the towers' PERSISTENT prompt-gradient buffers: gradients of vis_prompt / txt_prompt must not be retained (hooks,
    retain_grad) across this call, and no other forward may run on `enc` between this call's forward and backward (it does not: both are in here)."""

    def mark(name):
        if marks is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append((name, e))

    for k in PROMPT_NAMES:
        factors[k].grad = None
    gather = exchange.gather if exchange is not None else None
    gv = None if grad_views is None else [grad_views[list(factors).index(k)] for k in _CP_ORDER]
    _, img_f, txt_f, vis, txt = forward_loss(enc, images, ids, factors, depth, gather, align_weight, overlap_towers, vision_lanes, text_lanes,
                                             lockstep, exchange, grad_out=gv, losses=False)
    mark("forward")
    world = exchange.world if exchange is not None else 1
    w_base = float(getattr(exchange, "loss_weight", 1.0))
    # The alignment loss's gradient and the towers' gradient of the same prompt stacks are summed WITHOUT a sum kernel: the alignment kernel writes its
    # (dense, all-layer) gradient into the towers' persistent prompt-gradient buffers and the towers' backward adds its rows (layers < depth) on top
    # (DualEncoder.seed_prompt_grads).  The stacks then receive ONE gradient each, from the encoder node, so autograd has nothing to accumulate; the node
    # hands the buffers on without a copy ('dprompts_borrow': DecomposedPromptFn.backward consumes them at once).  Towers on lanes of their own
    # (overlap_towers) keep the plain path: the stacks are autograd roots too and autograd adds the two gradients.
    seed = not overlap_towers
    if task_term is not None and not seed:
        raise ValueError("task_term needs the seeded prompt-gradient path (the towers on one stream)")
    # the contexts of THIS call's forward (EncodeBothFn keeps them on its node; the lock-step / separate-node paths leave them on the engine)
    vis_ctx, txt_ctx = (enc._vis_ctx, enc._txt_ctx) if seed else (None, None)
    wss = [vis_ctx[0], txt_ctx[0]] if seed else []
    task_out = None
    try:      # the workspaces' seeding flags never outlive this step, whatever raises between the seeding and the backward
        with torch.no_grad():
            al = {}
            out_bufs = enc.seed_prompt_grads(vis_ctx, txt_ctx) if seed else None

            def align():
                # the alignment weight / W goes into the kernel: loss value and gradients come out scaled; the reported value is unscaled again below
                al["r"] = E.align_loss_fwd_bwd(vis.detach().float(), txt.detach().float(), 0.01, align_weight / world, True, out=out_bufs)

            if gather is not None and getattr(exchange, "device_collectives", False):
                # the data-independent alignment kernel runs while the feature all-gather (issued right behind the towers) is in flight
                g2 = lambda i, t: exchange.gather(i, t, between=align)  # noqa: E731
                base, dI, dT, _ = contrastive_loss_and_grads(img_f, txt_f, enc.logit_scale_exp, g2, exchange, True)
            else:
                base, dI, dT, _ = contrastive_loss_and_grads(img_f, txt_f, enc.logit_scale_exp, gather, exchange, True)
                align()
            align, dv, dt = al["r"]
            if task_term is not None:      # after the alignment kernel has WRITTEN the buffers: the task term adds onto them
                task_out = task_term(vis.detach(), txt.detach(), out_bufs[0], out_bufs[1], 1.0 / world)
            if w_base != 1.0:
                dI, dT = dI * w_base, dT * w_base
        mark("losses")
        if seed:
            if vis_ctx[8] != enc.vis.serial or txt_ctx[8] != enc.txt.serial:
                raise RuntimeError("train_step: another forward ran on this engine between the step's forward and its backward")
            for w in wss:
                w["dprompts_borrow"] = True
            torch.autograd.backward([img_f, txt_f], [dI, dT])
        else:
            torch.autograd.backward([img_f, txt_f, vis, txt], [dI, dT, dv, dt])
    finally:
        for w in wss:
            w.pop("dprompts_borrow", None)
            w.pop("dprompts_seeded", None)
    mark("backward")
    if exchange is not None:
        if gv is not None:
            exchange.allreduce_grads([factors[k] for k in PROMPT_NAMES], flat=flat_grad)
        else:
            exchange.allreduce_grads([factors[k] for k in PROMPT_NAMES])
    align_out = align[0] if world == 1 else align[0] * float(world)
    out = {"img_f": img_f.detach(), "txt_f": txt_f.detach(), "vis_prompt": vis.detach(), "txt_prompt": txt.detach(),
           "base_loss": base[0], "alignment_loss": align_out}
    if task_out is not None:
        out["task_loss"] = task_out
    return out

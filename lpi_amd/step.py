"""One training step of the LPI hot path on the HIP engine, written the way the reference's hot loop runs it
(methods/sprompt.py:297-311: forward -> cal_loss -> sum of losses -> backward), minus the optimiser.

Used by bench.py, __graft_entry__.smoke() and the parity tests; the plugin surface (lpi_amd/retrieval) reaches the
same Functions through SliNet.forward / SliNet.cal_loss.
"""
from __future__ import annotations

import torch

from .functional import AlignLossFn, ClipLossFn, DecomposedPromptFn, EncodeImageFn, EncodeTextFn
from .synth import PROMPT_NAMES


def _side_stream(enc):
    st = getattr(enc, "_side_stream", None)
    if st is None:
        st = enc._side_stream = torch.cuda.Stream(device=enc.device)
    return st


def forward_loss(enc, images, ids, factors: dict, depth: int = 1, gather=None, align_weight: float = 0.1, overlap_towers: bool = True):
    """factors: the five DecomposedPrompt parameters (device tensors, requires_grad as wanted).
    Returns (losses dict of 0-d tensors, img_f, txt_f, vis, txt, logits)."""
    vis, txt = DecomposedPromptFn.apply(*[factors[k] for k in ("dim_1_share", "dim_2_visual", "dim_2_textual", "dim_3_visual", "dim_3_textual")])
    if overlap_towers:
        # The towers are independent until the similarity matrix: run the text tower on a second HIP stream so its kernels fill
        # the CUs the vision tower's last partial round of GEMM tiles leaves idle (autograd replays each node's backward on the
        # stream its forward ran on, so the backward overlaps the same way).
        main = torch.cuda.current_stream()
        side = _side_stream(enc)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            txt_f = EncodeTextFn.apply(enc, ids, txt, depth)
        img_f = EncodeImageFn.apply(enc, images, vis, depth)
        main.wait_stream(side)
        txt_f.record_stream(main)
    else:
        img_f = EncodeImageFn.apply(enc, images, vis, depth)
        txt_f = EncodeTextFn.apply(enc, ids, txt, depth)
    fn = ClipLossFn
    base = fn.apply(img_f, txt_f, enc.logit_scale_exp, gather)
    losses = {"base_loss": base, "alignment_loss": AlignLossFn.apply(vis, txt, 0.01, align_weight)}
    return losses, img_f, txt_f, vis, txt


def train_step(enc, images, ids, factors: dict, depth: int = 1, exchange=None, align_weight: float = 0.1, overlap_towers: bool = True):
    """forward + losses + backward; leaves the gradients in factors[k].grad and returns the forward outputs.

    exchange: a ``dp.Exchange`` for data-parallel runs (features all-gathered for the global contrastive matrix, factor
    gradients SUM all-reduced; the data-independent alignment term is scaled by 1/W so that it counts once)."""
    for k in PROMPT_NAMES:
        factors[k].grad = None
    gather = exchange.gather if exchange is not None else None
    losses, img_f, txt_f, vis, txt = forward_loss(enc, images, ids, factors, depth, gather, align_weight, overlap_towers)
    world = exchange.world if exchange is not None else 1
    total = losses["base_loss"] + losses["alignment_loss"] / world
    total.backward()
    if exchange is not None:
        exchange.allreduce_grads([factors[k] for k in PROMPT_NAMES])
    return {"img_f": img_f.detach(), "txt_f": txt_f.detach(), "vis_prompt": vis.detach(), "txt_prompt": txt.detach(),
            "base_loss": losses["base_loss"].detach(), "alignment_loss": losses["alignment_loss"].detach()}

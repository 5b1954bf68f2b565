"""Data-parallel exchange for the contrastive step: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" in the CPU plumbing tests).

The reference has no working multi-GPU path (README.md:13); its dead ``gather_features``
(methods/sprompt.py:38-82) is the specification followed here with ``local_loss=False, gather_with_grad=False``:
every rank all-gathers the L2-normalised features, evaluates the FULL global loss, and back-propagates only through
its own rows; the prompt-factor gradients are then SUM all-reduced.  Both messages are tiny and latency bound
(1 MB and 21 KB at B=256), so each is ONE fused collective: image||text features in one all-gather, the five factor
gradients in one flat all-reduce (reduced after the CP contraction: 5 284 floats, not the 184 K dense ones).
Data-independent loss terms (alignment / task loss) are identical on every rank and must count once: the step scales
them by 1/W before the SUM.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class Exchange:
    def __init__(self, group=None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def gather(self, img_f: torch.Tensor, txt_f: torch.Tensor):
        """-> (img_all [W*B,E], txt_all [W*B,E], first global row of this rank)."""
        B, E = img_f.shape
        local = torch.cat([img_f, txt_f], dim=1).contiguous()
        out = torch.empty(self.world * B, 2 * E, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=self.group)
        return out[:, :E].contiguous(), out[:, E:].contiguous(), self.rank * B

    def allreduce_grads(self, params):
        """SUM all-reduce of the (small) prompt-factor gradients as one flat message."""
        params = [p for p in params if p.grad is not None]
        flat = torch.cat([p.grad.reshape(-1) for p in params])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        o = 0
        for p in params:
            n = p.grad.numel()
            p.grad.copy_(flat[o:o + n].view_as(p.grad))
            o += n
        return flat.numel()

"""ClipLoss / nt_bxent_loss with the reference's call signatures (loss/loss.py:6-87).

ClipLoss.forward(logits) takes a PRE-SCALED logits matrix, as in the reference; SliNet.cal_loss here does not go through it
for the base loss (it fuses logits + loss in the HIP ClipLossFn) but it is kept for callers that hold a logits matrix.
nt_bxent_loss (task loss, only when numtask != 1) runs in torch ops on <= 12 x 12 problems — SURVEY.md section 8(f) item 3,
not part of the measured path."""
import torch
from torch import nn
from torch.nn import functional as F


def nt_bxent_loss(x, target, temperature=1.0):
    """loss.py:6-33, including its sigmoid-then-BCE-with-logits as written."""
    n = x.size(0)
    target = target.type(torch.float32).to(x.device)
    xcs = F.cosine_similarity(x[None, :, :], x[:, None, :], dim=-1)
    xcs = xcs.masked_fill(torch.eye(n, dtype=torch.bool, device=x.device), float("inf"))
    loss = F.binary_cross_entropy_with_logits((xcs / temperature).sigmoid(), target, reduction="none")
    pos = target.bool()
    loss_pos = torch.where(pos, loss, torch.zeros_like(loss)).sum(1)
    loss_neg = torch.where(~pos, loss, torch.zeros_like(loss)).sum(1)
    num_pos = target.sum(1)
    return (loss_pos / num_pos + loss_neg / (n - num_pos)).mean()


class _ClipLossOnLogitsFn(torch.autograd.Function):
    """Symmetric CE of a pre-scaled [n, n] logits matrix through lpi_clip_loss_fwd_bwd (row / column log-sum-exp kernels)."""

    @staticmethod
    def forward(ctx, lg):
        from lpi_amd import engine as E
        n = lg.shape[0]
        lgc = lg.detach().float().contiguous()
        loss = torch.zeros(1, device=lg.device)
        dl = torch.zeros_like(lgc)
        lse = torch.zeros(2, n, device=lg.device)
        E.call("lpi_clip_loss_fwd_bwd", n, lgc, n, 1.0, loss, dl, n, lse[0], lse[1], E._stream())
        ctx.save_for_backward(dl)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        return ctx.saved_tensors[0] * g


class ClipLoss(nn.Module):
    """Same constructor keywords as the reference's ClipLoss (loss/loss.py:38-53); forward(logits) as loss.py:75-87."""

    def __init__(self, local_loss=False, gather_with_grad=False, cache_labels=True, rank=0, world_size=1, use_horovod=False):
        super().__init__()
        self.local_loss, self.gather_with_grad, self.cache_labels = local_loss, gather_with_grad, cache_labels
        self.rank, self.world_size, self.use_horovod = rank, world_size, use_horovod

    def forward(self, logits):
        return _ClipLossOnLogitsFn.apply(logits)

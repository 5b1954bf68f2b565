"""ClipLoss / nt_bxent_loss with the reference's call signatures (loss/loss.py:6-87), both on the HIP kernels.

ClipLoss.forward(logits) takes a PRE-SCALED logits matrix, as in the reference; SliNet.cal_loss here does not go through it
for the base loss (it fuses logits + loss in the HIP ClipLossFn) but it is kept for callers that hold a logits matrix.
nt_bxent_loss (task loss, only when numtask != 1) runs in lpi_nt_bxent_fwd_bwd.  There is no torch-op / CPU form of either."""
import torch
from torch import nn


def nt_bxent_loss(x, target, temperature=1.0, row=None):
    """loss.py:6-33 (including its sigmoid-then-BCE-with-logits as written) through lpi_nt_bxent_fwd_bwd.  x: [T, D] on the GPU —
    the stacked, flattened prompts of tasks 0..T-1 (slinet.py:179-181).  Only row `row` (default: the last = the task being trained;
    the other tasks' prompts are frozen, sprompt.py:230-237) receives a gradient."""
    from lpi_amd import _lib
    from lpi_amd.functional import NtBxentFn
    if not x.is_cuda:
        raise _lib.LpiError("nt_bxent_loss runs on the MI355X HIP kernels only (x must be a cuda tensor); there is no CPU fallback")
    return NtBxentFn.apply(x, target, float(temperature), x.size(0) - 1 if row is None else int(row))


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

"""Data-parallel exchange for the contrastive step: one process per GPU, torch.distributed (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" in the CPU plumbing tests and in the two-processes-on-one-GPU test of the HIP path).

The reference has no working multi-GPU path (README.md:13); its dead ``gather_features``
(methods/sprompt.py:38-82) is the specification followed here with ``local_loss=False, gather_with_grad=False``:
every rank all-gathers the L2-normalised features, evaluates the FULL global loss, and back-propagates only through
its own rows (engine.clip_loss_fwd_bwd computes the local rows of dlogits / dlogits^T only); the prompt-factor
gradients are then SUM all-reduced.  Both messages are tiny and latency bound (1 MB and 21 KB at B=256), so each is
ONE fused collective: image||text features in one all-gather, the five factor gradients in one flat all-reduce
(reduced after the CP contraction: 5 284 floats, not the 184 K dense ones).  Data-independent loss terms (alignment /
task loss) are identical on every rank and must count once: the step scales them by 1/W before the SUM.

The gathered [W*B, 2E] buffer is handed to the loss as two row-strided views (no unpacking copies).  With a backend that
cannot move device memory (gloo) the two messages are staged through the host.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def all_gather_rows(t: torch.Tensor, group=None) -> torch.Tensor:
    """Concatenation over ranks of the row blocks t [n_r, E] (n_r may differ per rank: the last shard of a DistributedSampler without drop_last, or
    shards of a dataset whose size is not a multiple of W).  Two collectives: the row counts, then ONE all_gather_into_tensor of the blocks padded to the
    largest — device to device on RCCL; staged through the host (as tensors, not pickles) where the backend cannot move device memory.  Used by the
    task-key clustering (methods/sprompt.py:370-397 under data parallelism: every rank clusters the features of all shards)."""
    world = dist.get_world_size(group)
    if world == 1:
        return t
    # the staging device follows the BACKEND, not the tensor: RCCL moves device memory only (a host tensor — the scikit-learn clustering keeps its features on
    # the host — goes through the current device), gloo moves host memory only.  The result comes back where `t` lives.
    if dist.get_backend(group) == "nccl":
        w = t if t.is_cuda else t.to(torch.device("cuda", torch.cuda.current_device()))
    else:
        w = t.cpu() if t.is_cuda else t
    counts = torch.zeros(world, dtype=torch.int64, device=w.device)
    mine = torch.tensor([t.shape[0]], dtype=torch.int64, device=w.device)
    dist.all_gather_into_tensor(counts, mine, group=group)
    counts = [int(c) for c in counts.cpu()]
    nmax = max(counts)
    pad = torch.zeros(nmax, t.shape[1], dtype=w.dtype, device=w.device)
    pad[:t.shape[0]].copy_(w)
    out = torch.empty(world * nmax, t.shape[1], dtype=w.dtype, device=w.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    rows = torch.cat([out[r * nmax:r * nmax + c] for r, c in enumerate(counts)], 0)
    return rows.to(t.device)


class Exchange:
    def __init__(self, group=None, timing: bool = False, local_loss: bool = False, gather_with_grad: bool = False):
        """local_loss / gather_with_grad: the modes of the reference's gather_features / get_logits (sprompt.py:38-82, 272-288),
        see functional.ClipLossFn.  The default (False, False) needs no backward collective."""
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.local_loss, self.gather_with_grad = bool(local_loss), bool(gather_with_grad)
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device_collectives = dist.get_backend(group) == "nccl"
        self.timing = [] if timing else None      # (kind, start event, end event) per collective, for bench.py
        self._local = {}

    def _run(self, kind, issue, on_device, between=None):
        """Issue one collective and make the CURRENT stream wait for it.  `issue()` returns the c10d Work of an async_op=True call; RCCL runs it on
        its own stream, ordered behind the current stream's work by an event at issue time.  `between()` (optional) is enqueued on the current
        stream after the issue and before the wait: it runs UNDER the collective.  Timing (bench.py): the start event is recorded on the current
        stream right before the issue, the end event right after Work.wait() has put the current stream behind the collective's own end event — the
        pair brackets the collective itself (stream-ordered through the two waits), i.e. the time the compute stream is held, whichever stream
        the transport used; a `between` kernel is inside the bracket and is named in `kind`."""
        t = self.timing is not None and on_device
        if t:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        work = issue()
        if between is not None:
            between()
        if work is not None:
            work.wait()     # the current stream waits for the collective; the host does not (device tensors)
        if t:
            e1.record()
            self.timing.append((kind, e0, e1))

    def _buf(self, name, shape, like):
        """A persistent message buffer (one per name / shape / dtype / device): collectives never allocate or `cat` in the step."""
        key = (name, tuple(shape), like.dtype, like.device)
        b = self._local.get(key)
        if b is None:
            b = self._local[key] = torch.empty(*shape, dtype=like.dtype, device=like.device)
        return b

    @staticmethod
    def _pack2(dst, a, b):
        """dst[:, :E] = a, dst[:, E:] = b  (lpi_copy_rows on the device: no ATen kernel on the step; plain copies on the host)."""
        E = a.shape[1]
        if dst.is_cuda and dst.dtype == torch.float32 and a.dtype == torch.float32 and b.dtype == torch.float32:
            from . import _lib
            s = torch.cuda.current_stream().cuda_stream
            a, b = (a if a.stride(1) == 1 else a.contiguous()), (b if b.stride(1) == 1 else b.contiguous())
            _lib.call("lpi_copy_rows", a.shape[0], E, a, a.stride(0), dst, 2 * E, s)
            _lib.call("lpi_copy_rows", b.shape[0], E, b, b.stride(0), dst[:, E:], 2 * E, s)
        else:
            dst[:, :E].copy_(a)
            dst[:, E:].copy_(b)

    def gather(self, img_f: torch.Tensor, txt_f: torch.Tensor, between=None):
        """-> (img_all [W*B,E], txt_all [W*B,E] (views of one [W*B, 2E] buffer), first global row of this rank).
        between: a callable run after the all-gather has been ISSUED and before its result is waited for (step.train_step puts the
        data-independent alignment-loss kernel there, so that it runs under the collective)."""
        B, E = img_f.shape
        local = self._buf("gather.local", (B, 2 * E), img_f)
        self._pack2(local, img_f, txt_f)
        if self.device_collectives or not local.is_cuda:
            # a fresh output per step: the views handed on are saved by the loss for its backward
            out = torch.empty(self.world * B, 2 * E, dtype=local.dtype, device=local.device)
            kind = "all_gather" if between is None or not local.is_cuda else "all_gather (alignment-loss kernel under it)"
            self._run(kind, lambda: dist.all_gather_into_tensor(out, local, group=self.group, async_op=True), local.is_cuda,
                      between if local.is_cuda else None)
            if local.is_cuda:
                between = None
        else:       # host-staged (gloo with device tensors)
            h = local.cpu()
            oh = torch.empty(self.world * B, 2 * E, dtype=h.dtype)
            dist.all_gather_into_tensor(oh, h, group=self.group)
            out = oh.to(local.device)
        if between is not None:
            between()
        return out[:, :E], out[:, E:], self.rank * B

    @property
    def loss_weight(self) -> float:
        """Weight of a rank's contrastive loss under the SUM all-reduce of the parameter gradients: 1 in the default mode (every rank
        holds the global loss and its own rows' share of the gradient), 1/W when every rank holds a full gradient of its own loss."""
        return 1.0 / self.world if (self.local_loss or self.gather_with_grad) else 1.0

    def reduce_scatter_rows(self, dI_all: torch.Tensor, dT_all: torch.Tensor, B: int):
        """SUM over ranks of the [W*B, E] key gradients, this rank's B rows of it: the backward of torch.distributed.nn.all_gather
        (sprompt.py:67-69).  One fused message (image || text); reduce_scatter on RCCL, all_reduce + slice where the backend has none."""
        E = dI_all.shape[1]
        buf = self._buf("rs.in", (dI_all.shape[0], 2 * E), dI_all)
        self._pack2(buf, dI_all, dT_all)
        if self.device_collectives and buf.is_cuda:      # RCCL, any world size (W = 1 included: the same call the 8-GPU run makes)
            out = torch.empty(B, 2 * E, dtype=buf.dtype, device=buf.device)
            self._run("reduce_scatter", lambda: dist.reduce_scatter_tensor(out, buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True), True)
        elif self.world == 1:
            out = buf.clone()
        else:
            h = buf.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            out = h[self.rank * B:(self.rank + 1) * B].to(buf.device)
        return out[:, :E], out[:, E:]

    def allreduce_grads(self, params, flat=None):
        """SUM all-reduce of the (small) prompt-factor gradients as one flat message.  flat: the flat gradient vector the parameters' .grad
        tensors are slices of (optim.flatten): reduced in place, nothing packed or copied back."""
        if flat is not None and all(p.grad is not None and p.grad.untyped_storage().data_ptr() == flat.untyped_storage().data_ptr() for p in params):
            if self.device_collectives or not flat.is_cuda:
                self._run("all_reduce", lambda: dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True), flat.is_cuda)
            else:
                h = flat.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                flat.copy_(h)
            return flat.numel()
        params = [p for p in params if p.grad is not None]
        flat = torch.cat([p.grad.reshape(-1) for p in params])
        if self.device_collectives or not flat.is_cuda:
            self._run("all_reduce", lambda: dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True), flat.is_cuda)
        else:
            h = flat.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            flat = h.to(flat.device)
        o = 0
        for p in params:
            n = p.grad.numel()
            p.grad.copy_(flat[o:o + n].view_as(p.grad))
            o += n
        return flat.numel()

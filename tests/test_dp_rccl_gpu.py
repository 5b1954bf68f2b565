"""The DEVICE-collective branch of lpi_amd/dp.py on RCCL (torch.distributed backend "nccl"), which the gloo tests never take:
`all_gather_into_tensor(async_op=True)` with the alignment-loss kernel issued under it and `Work.wait()`, `reduce_scatter_tensor`, the flat
in-place `all_reduce` — the calls an 8-GPU run makes (spec: the reference's dead gather_features / get_logits, sprompt.py:38-82, 272-288).

A one-GPU box has one rank to give RCCL (two ranks cannot share a device under RCCL), so the group has world size 1: every collective is
then the identity on the data, which makes the check sharp — the step through RCCL must equal, BIT FOR BIT, the same step with an exchange
object that moves nothing, in all four modes of gather_features; and in the three modes whose gradients are complete it must equal the oracle
on the same batch.  The group is created in a fresh child process before anything there touches the GPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from lpi_amd import synth  # noqa: E402

B, DEPTH = 6, 2
MODES = [(False, False), (True, False), (True, True), (False, True)]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _NoComm:
    """The interface step.train_step / functional.contrastive_loss_and_grads use of dp.Exchange, for ONE rank, moving nothing."""

    def __init__(self, local_loss, gather_with_grad):
        self.local_loss, self.gather_with_grad = local_loss, gather_with_grad
        self.world, self.rank, self.device_collectives = 1, 0, False

    @property
    def loss_weight(self):
        return 1.0

    def gather(self, img_f, txt_f, between=None):
        if between is not None:
            between()
        return img_f.contiguous(), txt_f.contiguous(), 0

    def reduce_scatter_rows(self, dI_all, dT_all, B):
        return dI_all[:B].clone(), dT_all[:B].clone()

    def allreduce_grads(self, params, flat=None):
        return 0


def _worker(port, q):
    try:
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        from lpi_amd import _lib
        from lpi_amd.dp import Exchange
        from lpi_amd.engine import DualEncoder
        from lpi_amd.optim import flatten
        from lpi_amd.step import train_step
        cfg = synth.TINY
        enc = DualEncoder(cfg, synth.clip_state_dict(cfg), dtype="f32", device=dev)
        img = torch.from_numpy(synth.images(B, cfg.image_resolution)).to(dev)
        ids = torch.from_numpy(synth.token_ids(B)).to(dev)
        res = {}
        for ll, gwg in MODES:
            for name in ("rccl", "nocomm"):
                fac = {k: torch.from_numpy(v).to(dev).requires_grad_(True)
                       for k, v in synth.prompt_factors(9, 16, cfg.vision_width, cfg.transformer_width).items()}
                _, flat_grad, views = flatten(fac)
                ex = Exchange(timing=True, local_loss=ll, gather_with_grad=gwg) if name == "rccl" else _NoComm(ll, gwg)
                if name == "rccl":
                    assert ex.device_collectives and dist.get_backend() == "nccl" and ex.world == 1
                n0 = _lib.launch_count()
                out = train_step(enc, img, ids, fac, DEPTH, ex, flat_grad=flat_grad, grad_views=views)
                torch.cuda.synchronize()
                assert _lib.launch_count() - n0 > 30, "the HIP kernels did not run"
                kinds = None
                if name == "rccl":
                    kinds = [k for k, _, _ in ex.timing]
                    times = [e0.elapsed_time(e1) for _, e0, e1 in ex.timing]
                    assert all(0.0 < t < 1e3 for t in times), times          # milliseconds, stream-ordered brackets
                res[(ll, gwg, name)] = (float(out["base_loss"]), float(out["alignment_loss"]), flat_grad.cpu().numpy().copy(),
                                        {k: v.grad.cpu().numpy().copy() for k, v in fac.items()}, kinds)
        q.put(("ok", res))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:      # surface the failure in the parent instead of a queue timeout
        import traceback
        q.put(("err", repr(e) + "\n" + traceback.format_exc()))
        raise


def test_rccl_branch_of_the_exchange_every_gather_mode():
    from oracle import lpi_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker, args=(_free_port(), q))
    p.start()
    status, res = q.get(timeout=900)
    p.join(180)
    assert status == "ok", res
    assert p.exitcode == 0
    cfg = synth.TINY
    ref = O.train_step(O.Oracle(cfg, synth.clip_state_dict(cfg)), synth.images(B, cfg.image_resolution), synth.token_ids(B),
                       synth.prompt_factors(9, 16, cfg.vision_width, cfg.transformer_width), depth=DEPTH)
    for ll, gwg in MODES:
        base, align, flat, grads, kinds = res[(ll, gwg, "rccl")]
        base0, align0, flat0, grads0, _ = res[(ll, gwg, "nocomm")]
        # the collectives of a one-rank group are the identity: the same bits as the step that moves nothing
        assert base == base0 and align == align0, (ll, gwg)
        assert np.array_equal(flat, flat0), (ll, gwg)
        # which collectives ran, and that the all-gather had the alignment-loss kernel issued under it (Exchange.gather(between=...))
        assert kinds[0] == "all_gather (alignment-loss kernel under it)", kinds
        assert kinds[-1] == "all_reduce", kinds
        assert ("reduce_scatter" in kinds) == gwg, kinds
        assert abs(base - float(ref["base_loss"])) < 1e-4 and abs(align - float(ref["alignment_loss"])) < 1e-4
        if gwg or not ll:       # complete gradients (local_loss alone is the reference's partial gradient: sprompt.py:75-80 without the re-insert)
            for k, g in grads.items():
                r = ref["grad." + k]
                assert np.abs(g - r).max() <= 1e-3 * np.abs(r).max() + 1e-6, (ll, gwg, k)
    # the partial-gradient mode really is a different gradient (the mode flag reaches the loss kernels)
    assert not np.array_equal(res[(True, False, "rccl")][2], res[(False, False, "rccl")][2])


def test_all_gather_rows_stages_host_tensors_on_the_device_under_rccl(monkeypatch):
    """with kmeans_impl='sklearn' the clustering features live on the HOST; RCCL moves device memory only, so dp.all_gather_rows must stage
    them through the current device (it handed host tensors to all_gather_into_tensor and raised).  Two ranks cannot share a GPU under RCCL, so the
    collective itself is a stand-in that refuses host tensors exactly as RCCL does and plays rank 1's part; the staging logic under test is dp's own."""
    import torch.distributed as dist
    from lpi_amd import dp

    calls = []

    def fake_all_gather_into_tensor(out, inp, group=None):
        assert out.is_cuda and inp.is_cuda, "RCCL cannot move host memory"
        calls.append(tuple(inp.shape))
        n = inp.shape[0]
        out[:n].copy_(inp)
        if inp.dtype == torch.int64:
            out[n:] = 3                       # rank 1 holds 3 rows
        else:
            out[n:].fill_(7.0)

    monkeypatch.setattr(dist, "get_world_size", lambda group=None: 2)
    monkeypatch.setattr(dist, "get_backend", lambda group=None: "nccl")
    monkeypatch.setattr(dist, "all_gather_into_tensor", fake_all_gather_into_tensor)
    torch.cuda.set_device(0)
    t = torch.arange(20, dtype=torch.float32).view(5, 4)            # a HOST tensor, 5 rows on this rank
    out = dp.all_gather_rows(t)
    assert out.device.type == "cpu" and out.shape == (8, 4)
    assert torch.equal(out[:5], t) and bool((out[5:] == 7.0).all())
    assert calls == [(1,), (5, 4)]
    d = dp.all_gather_rows(t.cuda())                                # device input: stays on the device
    assert d.is_cuda and torch.equal(d.cpu(), out)

"""Data parallelism of the HIP path, multi-process: two ranks (two processes) on GPU 0, each running the HIP engine on its shard of the
batch, exchanging through lpi_amd/dp.py (gloo group; the two tiny messages are staged through the host because two ranks cannot share
one device under RCCL).  What must hold (spec: the reference's dead gather_features / get_logits, sprompt.py:38-82, 272-288, with
local_loss=False): every rank reports the GLOBAL loss, and the SUM-all-reduced factor gradients equal the oracle's gradients on the
concatenated batch in one process."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from lpi_amd import synth  # noqa: E402

W, B = 2, 3
W4 = 4          # the GPU box admits six processes on the card: the test process + 4 ranks (8 ranks are rehearsed on the CPU: tests/test_dp_gloo.py)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _device_collective_adapters(dist):
    """The three c10d calls of dp.Exchange's DEVICE-collective branch (the one RCCL takes) on a backend that cannot move device memory: the same
    signatures, device tensors in and out, the transport staged through the host.  With them the branch an 8-GPU run takes — device message buffers,
    async_op calls, reduce_scatter_tensor — runs at world size 2 on one GPU (RCCL itself gives one rank per device: tests/test_dp_rccl_gpu.py)."""
    real_ag, real_ar = dist.all_gather_into_tensor, dist.all_reduce

    def all_gather_into_tensor(out, inp, group=None, async_op=False):
        assert out.is_cuda and inp.is_cuda and async_op
        oh = torch.empty(out.shape, dtype=out.dtype)
        real_ag(oh, inp.cpu(), group=group)
        out.copy_(oh)

    def reduce_scatter_tensor(out, inp, op=None, group=None, async_op=False):
        assert out.is_cuda and inp.is_cuda and async_op and op == dist.ReduceOp.SUM
        h = inp.cpu()
        real_ar(h, op=dist.ReduceOp.SUM, group=group)
        n, r = out.shape[0], dist.get_rank(group)
        out.copy_(h[r * n:(r + 1) * n])

    def all_reduce(t, op=None, group=None, async_op=False):
        if not t.is_cuda:
            return real_ar(t, op=op, group=group, async_op=async_op)
        h = t.cpu()
        real_ar(h, op=op, group=group)
        t.copy_(h)

    return all_gather_into_tensor, reduce_scatter_tensor, all_reduce


def _worker(rank, port, q, dtype, local_loss=False, gather_with_grad=False, W=W, flat=False, device_branch=False):
    try:
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(W))
        dist.init_process_group("gloo", rank=rank, world_size=W)
        from lpi_amd import _lib
        from lpi_amd.dp import Exchange
        from lpi_amd.engine import DualEncoder
        from lpi_amd.step import train_step
        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        cfg = synth.TINY
        enc = DualEncoder(cfg, synth.clip_state_dict(cfg), dtype=dtype, device=dev)
        fac = {k: torch.from_numpy(v).to(dev).requires_grad_(True)
               for k, v in synth.prompt_factors(9, 16, cfg.vision_width, cfg.transformer_width).items()}
        img = torch.from_numpy(synth.images(W * B, cfg.image_resolution))[rank * B:(rank + 1) * B].to(dev)
        ids = torch.from_numpy(synth.token_ids(W * B))[rank * B:(rank + 1) * B].to(dev)
        n0 = _lib.launch_count()
        kw = {}
        if flat:        # the factors in one flat vector, their gradients in another (optim.flatten): the all-reduce runs on it in place
            from lpi_amd.optim import flatten
            _, kw["flat_grad"], kw["grad_views"] = flatten(fac)
        ex = Exchange(local_loss=local_loss, gather_with_grad=gather_with_grad)
        if device_branch:
            import lpi_amd.dp as dp_mod
            ag, rs, ar = _device_collective_adapters(dist)
            dp_mod.dist.all_gather_into_tensor, dp_mod.dist.reduce_scatter_tensor, dp_mod.dist.all_reduce = ag, rs, ar
            ex.device_collectives = True
            first = train_step(enc, img, ids, fac, 2, ex, **kw)      # a first step: the second one reuses the persistent message buffers
            torch.cuda.synchronize()
            g1 = {k: v.grad.clone() for k, v in fac.items()}
        out = train_step(enc, img, ids, fac, 2, ex, **kw)
        torch.cuda.synchronize()
        if device_branch:
            assert float(first["base_loss"]) == float(out["base_loss"])
            assert all(torch.equal(g1[k], fac[k].grad) for k in fac), "the second step through the reused message buffers differs from the first"
        if flat:
            for v, p in zip(kw["grad_views"], fac.values()):
                assert p.grad.data_ptr() == v.data_ptr()
        assert _lib.launch_count() - n0 > 30, "the HIP kernels did not run"
        q.put((rank, float(out["base_loss"]), {k: v.grad.cpu().numpy().copy() for k, v in fac.items()}, out["img_f"].cpu().numpy()))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:      # surface the failure in the parent instead of a queue timeout
        q.put((rank, repr(e), None, None))
        raise


def test_two_process_hip_step_equals_oracle_on_global_batch():
    from oracle import lpi_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, q, "f32")) for r in range(W)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(W)), key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    cfg = synth.TINY
    ref = O.train_step(O.Oracle(cfg, synth.clip_state_dict(cfg)), synth.images(W * B, cfg.image_resolution), synth.token_ids(W * B),
                       synth.prompt_factors(9, 16, cfg.vision_width, cfg.transformer_width), depth=2)
    for rank, base, grads, img_f in res:
        assert grads is not None, base
        assert abs(base - float(ref["base_loss"])) < 1e-4            # every rank evaluates the full global loss
        assert np.abs(img_f - ref["img_f"][rank * B:(rank + 1) * B]).max() < 1e-4
        for k, g in grads.items():
            r = ref["grad." + k]
            assert np.abs(g - r).max() <= 1e-3 * np.abs(r).max() + 1e-6, (rank, k)
    for k in res[0][2]:      # both ranks hold the same (summed) gradients
        assert np.array_equal(res[0][2][k], res[1][2][k])


@pytest.mark.parametrize("local_loss,gather_with_grad", [(True, True), (False, True)])
def test_two_process_hip_step_with_gradients_through_the_gathered_features(local_loss, gather_with_grad):
    """gather_with_grad=True (sprompt.py:67-69): the key gradients are SUM reduce-scattered to their owners (host-staged under gloo), every
    rank weighs its loss by 1/W, and the SUM-all-reduced factor gradients are again the oracle's on the concatenated batch; with
    local_loss=True each rank reports its OWN mean loss (their mean is the global loss)."""
    from oracle import lpi_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, q, "f32", local_loss, gather_with_grad)) for r in range(W)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(W)), key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    cfg = synth.TINY
    ref = O.train_step(O.Oracle(cfg, synth.clip_state_dict(cfg)), synth.images(W * B, cfg.image_resolution), synth.token_ids(W * B),
                       synth.prompt_factors(9, 16, cfg.vision_width, cfg.transformer_width), depth=2)
    for rank, base, grads, img_f in res:
        assert grads is not None, base
        for k, g in grads.items():
            r = ref["grad." + k]
            assert np.abs(g - r).max() <= 1e-3 * np.abs(r).max() + 1e-6, (rank, k)
    mean_loss = float(np.mean([b for _, b, _, _ in res]))
    assert abs(mean_loss - float(ref["base_loss"])) < 1e-4
    if not local_loss:
        assert all(abs(b - float(ref["base_loss"])) < 1e-4 for _, b, _, _ in res)


@pytest.mark.parametrize("local_loss,gather_with_grad", [(False, False), (True, True), (False, True)])
def test_two_process_device_collective_branch_equals_oracle(local_loss, gather_with_grad):
    """The DEVICE-collective branch of dp.Exchange (device message buffers, async_op calls, all_gather_into_tensor / reduce_scatter_tensor / in-place
    all_reduce on device tensors — what an N-GPU RCCL run executes) at WORLD SIZE 2: two processes on GPU 0 with the three c10d calls replaced by
    host-staging adapters of the same signatures (_device_collective_adapters).  Two steps (the second reuses the persistent buffers) with flat
    gradients; the summed factor gradients equal the oracle's on the concatenated batch, in the modes whose gradients are complete."""
    from oracle import lpi_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, q, "f32", local_loss, gather_with_grad, W, True, True)) for r in range(W)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(W)), key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    cfg = synth.TINY
    ref = O.train_step(O.Oracle(cfg, synth.clip_state_dict(cfg)), synth.images(W * B, cfg.image_resolution), synth.token_ids(W * B),
                       synth.prompt_factors(9, 16, cfg.vision_width, cfg.transformer_width), depth=2)
    for rank, base, grads, img_f in res:
        assert grads is not None, base
        for k, g in grads.items():
            r = ref["grad." + k]
            assert np.abs(g - r).max() <= 1e-3 * np.abs(r).max() + 1e-6, (rank, k)
    assert abs(float(np.mean([b for _, b, _, _ in res])) - float(ref["base_loss"])) < 1e-4
    for k in res[0][2]:
        assert np.array_equal(res[0][2][k], res[1][2][k])


@pytest.mark.parametrize("local_loss,gather_with_grad", [(False, False), (True, False), (True, True), (False, True)])
def test_four_process_hip_step_every_gather_mode_flat_gradients(local_loss, gather_with_grad):
    """Four ranks on one GPU (gloo, host-staged messages), every mode of the reference's gather_features (sprompt.py:38-82) incl. the
    reduce-scatter path, with the factor gradients laid out in one flat vector that the all-reduce reduces in place.  With gradients
    through the gathered features, and in the default mode, the summed factor gradients are the oracle's on the concatenated batch; with
    local_loss alone they are the reference's partial gradient (checked against the single-process emulation in tests/test_kernels_gpu.py),
    here: every rank holds the same sum."""
    from oracle import lpi_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, q, "f32", local_loss, gather_with_grad, W4, True)) for r in range(W4)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(W4)), key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    cfg = synth.TINY
    ref = O.train_step(O.Oracle(cfg, synth.clip_state_dict(cfg)), synth.images(W4 * B, cfg.image_resolution), synth.token_ids(W4 * B),
                       synth.prompt_factors(9, 16, cfg.vision_width, cfg.transformer_width), depth=2)
    for rank, base, grads, img_f in res:
        assert grads is not None, base
        assert np.abs(img_f - ref["img_f"][rank * B:(rank + 1) * B]).max() < 1e-4
        if gather_with_grad or not local_loss:
            for k, g in grads.items():
                r = ref["grad." + k]
                assert np.abs(g - r).max() <= 1e-3 * np.abs(r).max() + 1e-6, (rank, k)
    for k in res[0][2]:
        for r in range(1, W4):
            assert np.array_equal(res[0][2][k], res[r][2][k])
    mean_loss = float(np.mean([b for _, b, _, _ in res]))
    assert abs(mean_loss - float(ref["base_loss"])) < 1e-4

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


def _worker(rank, port, q, dtype, local_loss=False, gather_with_grad=False, W=W, flat=False):
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
        out = train_step(enc, img, ids, fac, 2, Exchange(local_loss=local_loss, gather_with_grad=gather_with_grad), **kw)
        torch.cuda.synchronize()
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

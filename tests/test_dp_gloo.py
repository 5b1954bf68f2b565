"""Data-parallel exchange (lpi_amd/dp.py) with world_size 2 on CPU (gloo): W-rank result == single-process result on the
concatenated global batch.  The per-rank arithmetic is done by the ORACLE here (no GPU in this container); what is under test
is the exchange protocol: fused feature all-gather, local-rows-only gradient flow (``local_loss=False`` semantics of the
reference's gather_features, sprompt.py:38-82), 1/W scaling of data-independent terms, SUM all-reduce of factor grads."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lpi_amd import synth

B = 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, port, q, W):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(W))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    from lpi_amd.dp import Exchange
    from oracle import lpi_oracle as O
    cfg = synth.TINY
    orc = O.Oracle(cfg, synth.clip_state_dict(cfg))
    fac = {k: torch.from_numpy(v).requires_grad_(True) for k, v in synth.prompt_factors(9, 16, cfg.vision_width, cfg.transformer_width).items()}
    img = torch.from_numpy(synth.images(W * B, 32))[rank * B:(rank + 1) * B]
    ids = torch.from_numpy(synth.token_ids(W * B))[rank * B:(rank + 1) * B]
    ex = Exchange()
    img_f, txt_f, vp, tp = orc.forward(img, ids, fac, depth=2)
    ia, ta, r0 = ex.gather(img_f.detach(), txt_f.detach())
    assert r0 == rank * B and ia.shape == (W * B, cfg.embed_dim)
    ia = torch.cat([ia[:r0], img_f, ia[r0 + B:]])          # only this rank's rows carry gradient
    ta = torch.cat([ta[:r0], txt_f, ta[r0 + B:]])
    losses, _ = orc.cal_loss(ia, ta, vp, tp)
    (losses["base_loss"] + losses["alignment_loss"] / ex.world).backward()
    g0 = [v.grad.clone() for v in fac.values()]
    n = ex.allreduce_grads(list(fac.values()))
    assert n == sum(v.numel() for v in fac.values())
    # the same reduction with the gradients laid out as slices of ONE flat vector (optim.flatten): reduced in place, nothing packed or copied
    flat = torch.cat([g.reshape(-1) for g in g0])
    o = 0
    twins = [torch.zeros_like(v) for v in fac.values()]
    for t, g in zip(twins, g0):
        t.grad = flat[o:o + g.numel()].view_as(g)
        o += g.numel()
    assert ex.allreduce_grads(twins, flat=flat) == n
    for t, v in zip(twins, fac.values()):
        assert t.grad.data_ptr() >= flat.data_ptr() and torch.equal(t.grad, v.grad)
    q.put((rank, float(losses["base_loss"]), {k: v.grad.numpy().copy() for k, v in fac.items()}))
    dist.barrier()
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize("W", [2, 8])
def test_w_rank_step_equals_global_batch(W):
    """W = 8: the rank count of BASELINE configs[3] / [4] (8 GPUs), rehearsed as 8 CPU processes."""
    from oracle import lpi_oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, q, W)) for r in range(W)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(W)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    cfg = synth.TINY
    ref = O.train_step(O.Oracle(cfg, synth.clip_state_dict(cfg)), synth.images(W * B, 32), synth.token_ids(W * B),
                       synth.prompt_factors(9, 16, cfg.vision_width, cfg.transformer_width), depth=2)
    for rank, base, grads in res:
        assert abs(base - float(ref["base_loss"])) < 1e-5          # every rank evaluates the full global loss
        for k, g in grads.items():
            r = ref["grad." + k]
            assert np.abs(g - r).max() <= 1e-4 * np.abs(r).max() + 1e-7, (rank, k)
    # and all ranks end up with identical gradients
    for k in res[0][2]:
        for r in range(1, W):
            assert np.array_equal(res[0][2][k], res[r][2][k])


def _rows_worker(rank, port, q, W):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(W))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    from lpi_amd.dp import all_gather_rows
    n = 5 + 3 * rank                                     # ragged shards (the last shard of a sampler without drop_last)
    t = torch.arange(n * 4, dtype=torch.float32).view(n, 4) + 1000 * rank
    out = all_gather_rows(t)
    q.put((rank, out.numpy().copy(), str(out.device)))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_rows_of_host_tensors_with_ragged_shards():
    """dp.all_gather_rows on HOST tensors (the scikit-learn clustering keeps its features on the host: ADVICE round 5) with a different row count per rank:
    every rank receives the concatenation in rank order, on the device the input lives on."""
    W = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rows_worker, args=(r, port, q, W)) for r in range(W)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(W)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = np.concatenate([np.arange((5 + 3 * r) * 4, dtype=np.float32).reshape(-1, 4) + 1000 * r for r in range(W)])
    for rank, got, dev in res:
        assert dev == "cpu" and np.array_equal(got, want), rank

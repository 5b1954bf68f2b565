"""The plugin's hot loop under data parallelism (round 5): two processes on GPU 0 run SPrompts._setup_training + train_epoch on their shards — input pipeline,
fused SliNet.train_step WITH the task term of a second task, FlatSGD on the flat all-reduced gradient — and then the task-key clustering, which gathers the
ranks' features with dp.all_gather_rows.  What must hold (methods/sprompt.py:38-82 with local_loss=False; :370-397): after the epochs both ranks hold the
SAME parameters, equal to a single process training on the concatenated batches; the data-independent alignment and task terms counted once; and both ranks
end with the same KMeans keys as the single process."""
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from lpi_amd import synth  # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RET = os.path.join(REPO, "lpi_amd", "retrieval")
W, B, NB, EPOCHS = 2, 3, 2, 2          # ranks, pairs per rank and batch, batches per epoch, epochs


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _args(dev):
    args = json.load(open(os.path.join(RET, "configs", "lpi", "coco_lpi.json")))
    args.update(backbonename="tiny", visual_dim=128, textual_dim=128, device=[dev], compute_dtype="f32", batch_size=B, epochs=EPOCHS, num_workers=0)      # (batch_size is not read: the loaders are lists)
    return args


def _data():
    n = W * B * NB
    return torch.from_numpy(synth.images(n, 32, seed=77)), torch.from_numpy(synth.token_ids(n, seed=78))


def _run(rank, world):
    """Train task 2 (numtask = 2: the task term is live) on this rank's shard of every global batch; return the factors, the keys and the losses."""
    from lpi_amd.retrieval.methods.sprompt import LossLog, SPrompts
    dev = torch.device("cuda:0")
    m = SPrompts(_args(dev))
    net = m._network.to(dev)
    for t in range(len(net.prompts)):
        for k, v in synth.prompt_factors(9, 16, 128, 128, task=t).items():
            getattr(net.prompts[t], k).data = torch.from_numpy(v.copy()).to(dev)
    net.numtask = 2
    img, ids = _data()
    G = W * B                                         # the GLOBAL batch, the same for every world size; a rank takes its contiguous share of it
    share = G // world
    loader = []
    for b in range(img.shape[0] // G):
        lo = b * G + rank * share
        loader.append((img[lo:lo + share], ids[lo:lo + share], 0, 1))
    opt, sched = m._setup_training()
    log = LossLog()
    for ep in range(EPOCHS):
        m.train_epoch(loader, opt, ep, log)
        sched.step()
    # the gathered feature matrix is RANK-major (rank 0's batches, then rank 1's): k-means++ seeds by row index, so the one-process reference clusters
    # the same rows in that order
    cl = loader if world > 1 else [(img[b * G + r * B:b * G + (r + 1) * B], ids[b * G + r * B:b * G + (r + 1) * B], 0, 1)
                                   for r in range(W) for b in range(img.shape[0] // G)]
    m.clustering(cl)
    torch.cuda.synchronize()
    fac = {k: getattr(net.prompts[1], k).detach().cpu().numpy().copy() for k in synth.PROMPT_NAMES}
    return fac, m.all_keys[0].cpu().numpy().copy(), m.textual_all_keys[0].cpu().numpy().copy()


def _worker(rank, port, q):
    try:
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(W))
        dist.init_process_group("gloo", rank=rank, world_size=W)
        torch.cuda.set_device(0)
        out = _run(rank, W)
        q.put((rank,) + out)
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:      # surface the failure in the parent instead of a queue timeout
        q.put((rank, repr(e), None, None))
        raise


def test_two_rank_plugin_loop_equals_one_rank_on_the_concatenated_batches():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, port, q)) for r in range(W)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(W)), key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for r in res:
        assert r[2] is not None, r[1]
    ref_fac, ref_kv, ref_kt = _run(0, 1)              # one process, global batches of W * B pairs
    start = synth.prompt_factors(9, 16, 128, 128, task=1)
    for k in synth.PROMPT_NAMES:
        moved = np.abs(ref_fac[k] - start[k]).max()
        assert moved > 0
        assert np.array_equal(res[0][1][k], res[1][1][k]), k                      # the ranks agree bit for bit (same summed gradient, same update)
        assert np.abs(res[0][1][k] - ref_fac[k]).max() <= 2e-3 * moved + 1e-6, (k, np.abs(res[0][1][k] - ref_fac[k]).max(), moved)
    for r in res:                                      # every rank clustered the features of ALL shards
        for got, ref in ((r[2], ref_kv), (r[3], ref_kt)):
            assert got.shape == (5, 128)
            assert np.abs(got - ref).max() < 1e-5          # the same rows in the same (rank-major) order: the same fit
    assert np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][3], res[1][3])

"""Task keys on the device: the fit behind ``KMeans(n_clusters=5, random_state=0).fit(features)`` of the reference's ``clustering``
(retrieval/methods/sprompt.py:370-397) with the features left in HBM.

The algorithm is scikit-learn's (a dependency of the reference that is not under /root/reference; unpinned there, 1.7.2 in the build container):
``sklearn/cluster/_kmeans.py`` — KMeans.fit with init='k-means++', n_init='auto' (one run), algorithm='lloyd', max_iter=300, tol=1e-4.  What runs where:
  * HIP (csrc/loss.hip, lpi_kmeans_*): every pass over the features — squared distances to candidate centres, label assignment, centre means, the column
    statistics of the tolerance;
  * host: the random draws (numpy's RandomState, as scikit-learn uses it: the seeding must pick the SAME points), the cumulative sum / searchsorted of the
    seeding on a vector of n floats, and the convergence tests on k x E floats.  Per iteration a few kilobytes cross PCIe; the [n, E] features never do
    (the reference moves them to the host with .cpu().numpy(), sprompt.py:391-392).
The data are NOT centred here: scikit-learn centres them only to keep its |c|^2 - 2 x.c distance form accurate, and distances between points, label
assignments and centre means are the same on the raw data (the kernels take differences directly).
Pinned by tests/golden/kmeans.npz: the centres the imported reference's clustering() found (tests/test_round4_gpu.py, next to the CPU restatement the tests keep).
There is no CPU fallback: without the library or a GPU this raises.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from ._lib import call


def kmeans_fit(x: torch.Tensor, n_clusters: int = 5, random_state: int = 0, max_iter: int = 300, tol: float = 1e-4):
    """x: [n, E] f32 on the GPU -> (centers [k, E] f32 on the GPU, labels [n] int32 on the GPU, iterations)."""
    if not x.is_cuda:
        raise _lib.LpiError("lpi_amd.kmeans runs on an MI355X only (the features must be a cuda tensor); there is no CPU fallback")
    x = x.detach().float().contiguous()
    n, E = x.shape
    if n < n_clusters:
        raise ValueError(f"n_samples={n} should be >= n_clusters={n_clusters}.")      # scikit-learn's message
    dev = x.device
    s = torch.cuda.current_stream().cuda_stream
    k = n_clusters
    # tolerance: mean over the columns of their variance (np.var: about the column mean), times tol — two passes: the column means, then the squares of
    # the CENTRED values (E[x^2] - mean^2 from single-pass f32 sums cancels for columns whose mean dwarfs their deviation, and can go negative)
    colsum, colsq = torch.empty(E, device=dev), torch.empty(E, device=dev)
    call("lpi_kmeans_colstats", n, E, x, E, None, colsum, colsq, s)
    mean = (colsum.double() / n).float().contiguous()
    call("lpi_kmeans_colstats", n, E, x, E, mean, colsum, colsq, s)
    cs, cq = colsum.double().cpu().numpy(), colsq.double().cpu().numpy()
    tol_ = float(np.mean(np.maximum(cq / n - (cs / n) ** 2, 0.0))) * tol      # cs / n: the f32 rounding of the mean, a second-order correction
    # ---- k-means++ (sklearn _kmeans_plusplus): 2 + int(log k) local trials per centre
    rs = np.random.RandomState(random_state)
    w = np.ones(n, dtype=np.float32)
    trials = 2 + int(np.log(k))
    cand_dev = torch.empty(max(trials, 1), dtype=torch.int32, device=dev)
    dist_dev = torch.empty(max(trials, 1), n, device=dev)

    def sqdist(ids):
        m = len(ids)
        cand_dev[:m].copy_(torch.as_tensor(np.asarray(ids, dtype=np.int32)))
        call("lpi_kmeans_sqdist", n, E, m, x, E, cand_dev, dist_dev, s)
        return dist_dev[:m].cpu().numpy()

    idx = np.full(k, -1, dtype=np.int64)
    idx[0] = rs.choice(n, p=w / w.sum())
    closest = sqdist([idx[0]])
    pot = closest @ w
    for c in range(1, k):
        rand_vals = rs.uniform(size=trials) * pot
        cand = np.searchsorted(np.cumsum(w * closest, dtype=np.float64).ravel(), rand_vals.ravel())
        np.clip(cand, None, n - 1, out=cand)
        dc = sqdist(cand)
        np.minimum(closest, dc, out=dc)
        pots = dc @ w.reshape(-1, 1)
        best = int(np.argmin(pots))
        pot, closest = pots[best], dc[best:best + 1]
        idx[c] = cand[best]
    centers = x[torch.as_tensor(idx, device=dev)].contiguous()
    # ---- Lloyd iterations (sklearn _kmeans_single_lloyd)
    labels = torch.full((n,), -1, dtype=torch.int32, device=dev)
    changed = torch.zeros(1, dtype=torch.int32, device=dev)
    new = torch.empty_like(centers)
    counts = torch.empty(k, device=dev)
    mindist = torch.empty(n, device=dev)
    strict = False
    relocated = set()
    it = 0
    for it in range(max_iter):
        changed.zero_()
        call("lpi_kmeans_assign", n, E, k, x, E, centers, labels, changed, mindist, s)
        call("lpi_kmeans_update", n, E, k, x, E, labels, new, counts, s)
        ch = int(changed.item())                      # did any label move?  (4 bytes)
        new_h, old_h, cnt_h = new.cpu().numpy(), centers.cpu().numpy(), counts.cpu().numpy()      # k x E floats each: the convergence test runs on the host
        cycle = False
        if float(cnt_h.min()) == 0.0:
            new_h = _relocate_empty_clusters(x, labels, mindist, new_h, cnt_h)
            new.copy_(torch.from_numpy(new_h))
            # With fewer distinct rows than clusters every point coincides with a centre and "the farthest point" is decided by the last bit of a mean of
            # equal rows: the relocation can then hand one duplicate centre back and forth for ever (scikit-learn's own arithmetic happens to settle;
            # it would otherwise run to max_iter).  A relocation that reproduces centres already seen ends the fit: nothing new can come.
            key = new_h.tobytes()
            cycle = key in relocated
            relocated.add(key)
        shift2 = float((np.sqrt(((new_h - old_h) ** 2).sum(1)) ** 2).sum())
        centers, new = new, centers
        if not ch:
            strict = True
            break
        if shift2 <= tol_ or cycle:
            break
    if not strict:
        changed.zero_()
        call("lpi_kmeans_assign", n, E, k, x, E, centers, labels, changed, None, s)
    return centers, labels, it + 1


def _relocate_empty_clusters(x, labels, mindist, new_h, cnt_h):
    """scikit-learn's _relocate_empty_clusters_dense (sklearn/cluster/_k_means_common.pyx; called inside every Lloyd iteration before the sums are
    averaged): each cluster that lost all its points takes the point FARTHEST from its own (old) centre — the n_empty largest distances, in
    np.argpartition's order — and the donor cluster gives that point up.  The labels stay as they are (the next iteration re-assigns).  Rare (small or
    duplicate-heavy tasks: COCO repeats every image feature ~5 times), so it runs on the host on what the iteration already produced: the n distances,
    the labels, the k x E means; the donors' sums are rebuilt as mean x count in float64 (scikit-learn holds the float32 sums themselves: equal to
    rounding).  Returns the corrected means [k, E] float32."""
    empty = np.where(cnt_h == 0)[0]
    n_empty = empty.shape[0]
    d = mindist.cpu().numpy()
    far = np.argpartition(d, -n_empty)[:-n_empty - 1:-1]
    lab = labels[torch.as_tensor(far.copy(), device=labels.device)].cpu().numpy()
    pts = x[torch.as_tensor(far.copy(), device=x.device)].double().cpu().numpy()
    sums = new_h.astype(np.float64) * cnt_h.astype(np.float64)[:, None]
    cnt = cnt_h.astype(np.float64).copy()
    for j in range(n_empty):
        new_id, old_id = int(empty[j]), int(lab[j])
        sums[old_id] -= pts[j]
        sums[new_id] = pts[j]
        cnt[new_id] = 1.0
        cnt[old_id] -= 1.0
    safe = np.where(cnt > 0, cnt, 1.0)       # a donor emptied in turn keeps a zero centre, as _average_centers leaves it
    return (sums / safe[:, None]).astype(np.float32)

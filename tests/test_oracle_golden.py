"""Pin the oracle (oracle/lpi_oracle.py) against fixtures captured from the imported reference
(tests/golden/gen_golden.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from lpi_amd import synth
from oracle import lpi_oracle as O

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TASK_SIM = np.loadtxt(os.path.join(REPO, "lpi_amd", "retrieval", "MID", "task_sim_matrix.txt"))
GRADS = ["grad." + n for n in synth.PROMPT_NAMES]


def rel_err(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.fixture(scope="module")
def tiny():
    return O.Oracle(synth.TINY, synth.clip_state_dict(synth.TINY))


def factors(cfg, task=0):
    return synth.prompt_factors(9, 16, cfg.vision_width, cfg.transformer_width, task=task)


def check_step(res, g, tol=2e-5, gtol=1e-4):
    for k in ("img_f", "txt_f", "logits", "vis_prompt", "txt_prompt", "base_loss", "alignment_loss"):
        ref = g[k]
        got = res[k][: ref.shape[0]] if ref.ndim == 3 else res[k]
        assert np.abs(got - ref).max() <= tol * max(1.0, np.abs(ref).max()), k
    for k in GRADS:
        assert rel_err(res[k], g[k]) <= gtol, (k, rel_err(res[k], g[k]))


@pytest.mark.parametrize("name,depth", [("tiny_d1", 1), ("tiny_d2_patched", 2)])
def test_tiny_train_step(tiny, golden, name, depth):
    g = golden(name)
    res = O.train_step(tiny, synth.images(4, 32), g["token_ids"], factors(synth.TINY), depth=depth)
    check_step(res, g)


def test_tiny_depth_changes_result(golden):
    # the patched fixture must differ from the true-oracle one (otherwise the patch was a no-op)
    assert np.abs(golden("tiny_d1")["img_f"] - golden("tiny_d2_patched")["img_f"]).max() > 1e-4


def test_tiny_task_loss(tiny, golden):
    g = golden("tiny_task2")
    allf = [factors(synth.TINY, t) for t in range(2)]
    res = O.train_step(tiny, synth.images(4, 32), g["token_ids"], allf[1], depth=1, numtask=2,
                       all_factors_np=allf, task_sim=TASK_SIM)
    check_step(res, g)
    assert abs(res["task_loss"] - g["task_loss"]) <= 2e-5 * max(1.0, abs(g["task_loss"]))


def test_tiny_eval_interfaces(tiny, golden):
    g = golden("tiny_eval")
    cfg = synth.TINY
    img = torch.from_numpy(synth.images(6, 32, seed=synth.IMAGE_SEED + 7))
    ids = torch.from_numpy(g["token_ids"])
    allf = [{k: torch.from_numpy(v) for k, v in factors(cfg, t).items()} for t in range(12)]
    with torch.no_grad():
        ev = tiny.extract_vector(img)
        vi = tiny.visual_interface(img, torch.from_numpy(g["sel_v"]), allf)
        et = tiny.extract_textual_vector(ids)
        ti = tiny.textual_interface(ids, torch.from_numpy(g["sel_t"]), allf)
    assert np.abs(et.numpy() - g["extract_textual_vector"]).max() < 2e-5
    assert np.abs(ti.numpy() - g["textual_interface"]).max() < 2e-5
    assert (O.task_id_by_keys(et, torch.from_numpy(g["task_keys"])).numpy() == g["textual_task_id"]).all()
    with torch.no_grad():
        pass
    assert np.abs(ev.numpy() - g["extract_vector"]).max() < 2e-5
    assert np.abs(vi.numpy() - g["visual_interface"]).max() < 2e-5
    keys = torch.from_numpy(g["task_keys"])
    assert (O.task_id_by_keys(ev, keys).numpy() == g["visual_task_id"]).all()


def test_itm_eval(golden):
    g = golden("tiny_eval")
    s = g["itm_scores"]
    n_img, n_txt = s.shape
    fr = O.itm_eval(s, s.T.copy(), {t: t // 2 for t in range(n_txt)}, {i: [2 * i, 2 * i + 1] for i in range(n_img)},
                    g["itm_cat_i"], g["itm_cat_t"], 3)
    assert np.allclose([fr["mscoco"]["i2t"][t] for t in range(3)], g["itm_i2t"])
    assert np.allclose([fr["mscoco"]["t2i"][t] for t in range(3)], g["itm_t2i"])


@pytest.mark.parametrize("name,depth", [("vitb16_d1", 1), ("vitb16_d3_patched", 3)])
def test_vitb16_train_step(golden, name, depth):
    """BASELINE.json configs[0]: ViT-B/16, bs=8, r=4 on the CPU reference path."""
    cfg = synth.VIT_B16
    orc = O.Oracle(cfg, synth.clip_state_dict(cfg))
    g = golden(name)
    res = O.train_step(orc, synth.images(8, 224), g["token_ids"], factors(cfg), depth=depth)
    check_step(res, g, tol=5e-5, gtol=2e-3)
    # top-k index parity wherever the recorded margin dominates the logit error
    err = np.abs(res["logits"] - g["logits"]).max()
    for tag, S in (("i2t", res["logits"]), ("t2i", res["logits"].T)):
        idx = np.argsort(-S, axis=1, kind="stable")[:, : g[f"top5_{tag}"].shape[1]]
        safe = g[f"top5_margin_{tag}"] > 10 * err
        assert (idx[safe] == g[f"top5_{tag}"][safe]).all()


def test_text_columns_behind_every_eot_are_dead_in_the_oracle():
    """The property the engine's text trimming relies on (lpi_amd.engine.trim_token_ids), stated on the oracle that the fixtures pin
    to the reference: with the causal mask (model.py:347-353) and the EOT gather (prompt_learner.py:61), dropping the token columns
    behind the longest caption's EOT changes no feature, loss or factor gradient — here exactly (f64)."""
    import numpy as np
    import torch
    from lpi_amd import synth
    from lpi_amd.engine import trim_token_ids
    cfg = synth.TINY
    ora = O.Oracle(cfg, synth.clip_state_dict(cfg), dtype=torch.float64)
    ids = synth.token_ids(5, n_ctx=16, max_len=20)
    short = np.ascontiguousarray(trim_token_ids(ids))
    assert short.shape[1] == int(ids.argmax(-1).max()) + 1 < ids.shape[1]
    assert torch.equal(trim_token_ids(torch.from_numpy(ids)), torch.from_numpy(short))
    fac = synth.prompt_factors(9, 16, cfg.vision_width, cfg.transformer_width)
    img = synth.images(5, cfg.image_resolution)
    full = O.train_step(ora, img, ids, fac, depth=2)
    trim = O.train_step(ora, img, short, fac, depth=2)
    for k in full:
        assert float(np.abs(np.asarray(full[k], dtype=np.float64) - np.asarray(trim[k], dtype=np.float64)).max()) == 0.0, k



def test_tokens_behind_a_captions_own_eot_are_dead_in_the_oracle():
    """The property engine.PackedIds relies on: under the causal mask (model.py:347-353) and the EOT gather (prompt_learner.py:61) the
    positions behind a caption's OWN EOT can reach neither its feature nor a gradient — replacing the zero padding behind every EOT
    by arbitrary tokens (smaller than the EOT id, so ids.argmax(-1) still finds the EOT) changes nothing, exactly (f64); and
    PackedIds lays out exactly the live rows."""
    import numpy as np
    import torch
    from lpi_amd import synth
    from lpi_amd.engine import PackedIds
    cfg = synth.TINY
    ora = O.Oracle(cfg, synth.clip_state_dict(cfg), dtype=torch.float64)
    ids = synth.token_ids(5, n_ctx=16, max_len=20)
    eot = ids.argmax(-1)
    junk = ids.copy()
    rng = np.random.default_rng(0)
    for b in range(ids.shape[0]):
        junk[b, eot[b] + 1:] = rng.integers(1, 300, size=ids.shape[1] - eot[b] - 1)
    assert (junk.argmax(-1) == eot).all() and (junk != ids).any()
    fac = synth.prompt_factors(9, 16, cfg.vision_width, cfg.transformer_width)
    img = synth.images(5, cfg.image_resolution)
    a = O.train_step(ora, img, ids, fac, depth=2)
    b_ = O.train_step(ora, img, junk, fac, depth=2)
    for k in a:
        assert float(np.abs(np.asarray(a[k], dtype=np.float64) - np.asarray(b_[k], dtype=np.float64)).max()) == 0.0, k
    pk = PackedIds(ids)
    assert pk.rows == int((eot + 1).sum()) and pk.shape == (5, int(eot.max()) + 1)
    assert pk.row_start.tolist() == [0] + np.cumsum(eot + 1).tolist()
    assert pk.pool_rows.tolist() == (np.cumsum(eot + 1) - 1).tolist()
    assert torch.equal(pk.ids, torch.from_numpy(ids[:, :pk.shape[1]]))
    assert pk[1:3].rows == int((eot[1:3] + 1).sum())


def test_interact_module_oracle_vs_reference_fixture(golden):
    """SURVEY 8 (f4, optional): the oracle's restatement of the grounding branch's InteractModule against outputs and autograd gradients of the
    imported reference class (tests/golden/interact.npz; inputs regenerated from the seed)."""
    g = golden("interact")
    bs, P, Dv, Dt, layer_num, r = (int(x) for x in g["shape"])
    inp = synth.interact_inputs(bs, P, Dv, Dt)
    p = {k[6:]: torch.from_numpy(g[k]).double().requires_grad_(True) for k in g if k.startswith("param.")}
    v = torch.from_numpy(inp["visual_in"]).double().requires_grad_(True)
    t = torch.from_numpy(inp["textual_in"]).double().requires_grad_(True)
    vo, to = O.interact(p, v, t, int(g["layer_id"]))
    assert float((vo.detach() - torch.from_numpy(g["visual_out"])).abs().max()) < 2e-5
    assert float((to.detach() - torch.from_numpy(g["textual_out"])).abs().max()) < 2e-5
    ((vo * torch.from_numpy(inp["wv"])).sum() + (to * torch.from_numpy(inp["wt"])).sum()).backward()
    for name, got in [("visual_in", v.grad), ("textual_in", t.grad)] + [(k, p[k].grad) for k in p]:
        ref = torch.from_numpy(g["grad." + name]).double()
        assert float((got - ref).abs().max()) <= 1e-4 * float(ref.abs().max()) + 1e-7, name


def test_oracle_kmeans_equals_the_reference_clustering(golden):
    """oracle.kmeans_fit (a numpy restatement of scikit-learn's KMeans fit as the reference calls it, sprompt.py:393-394) against the centres the IMPORTED
    reference's clustering() found on the synthetic features (tests/golden/kmeans.npz, generated by tests/golden/gen_golden.py --only kmeans)."""
    import numpy as np
    from lpi_amd import synth
    from oracle import lpi_oracle as O
    g = golden("kmeans")
    n, dim = (int(x) for x in g["shape"])
    fv, ft = synth.clustering_features(n, dim)
    for name, f in (("visual", fv), ("textual", ft)):
        x = f / np.linalg.norm(f, axis=-1, keepdims=True)
        centers, labels, iters = O.kmeans_fit(x)
        assert np.abs(centers - g["centers_" + name]).max() < 1e-6, name
        assert labels.min() == 0 and labels.max() == 4 and iters < 50


def test_oracle_kmeans_relocates_empty_clusters_like_scikit_learn():
    """A feature set with fewer distinct rows (4) than clusters (5): k-means++ must seed one centre on a row it already chose, that cluster gets no point
    in the Lloyd iteration, and scikit-learn relocates it (_relocate_empty_clusters_dense) instead of failing.  The oracle's restatement against
    scikit-learn itself (the reference's own dependency, sprompt.py:393-394)."""
    import warnings
    import numpy as np
    from sklearn.cluster import KMeans
    from oracle import lpi_oracle as O
    from lpi_amd import synth
    x = synth.duplicate_heavy_features()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        km = KMeans(n_clusters=5, random_state=0).fit(x)
    centers, labels, _ = O.kmeans_fit(x)
    assert np.abs(centers - km.cluster_centers_).max() < 1e-5
    assert np.array_equal(labels, km.labels_)


@pytest.mark.parametrize("fp16,f32", [("tiny_fp16", "tiny_d1"), ("vitb16_fp16", "vitb16_d1")])
def test_fp16_family_is_the_reference_in_its_own_type(golden, fp16, f32):
    """tests/golden/*_fp16.npz: the imported reference run the way it ships — build_model's convert_weights (model.py:394-415, 522), SliNet.dtype fp16
    (slinet.py:30) — on torch's CPU fp16 kernels, same weights / inputs / captions as the f32 family.  Sanity of the fixture: it is the f32 family within
    fp16 rounding (and NOT equal to it), so it can pin the build's compute_dtype='f16' mode (tests/test_round5_gpu.py)."""
    a, b = golden(fp16), golden(f32)
    assert (a["token_ids"] == b["token_ids"]).all() and str(a["reference_dtype"]).startswith("float16")
    for k, bar in (("img_f", 1e-3), ("txt_f", 1e-3), ("logits", 1e-2)):
        e = float(np.abs(a[k] - b[k]).max())
        assert 0 < e < bar, (k, e)
    for k in ("base_loss", "alignment_loss"):
        assert abs(float(a[k]) - float(b[k])) < 2e-3 * max(1.0, abs(float(b[k])))
    for k in [k for k in a if k.startswith("grad.")]:
        cos = float((a[k] * b[k]).sum() / np.sqrt((a[k] ** 2).sum() * (b[k] ** 2).sum()))
        assert cos > 0.999, (k, cos)


# ----------------------------------------------------------------------------------------------- round 6: the fixtures at operating size
@pytest.mark.parametrize("fam", ["random", "drift", "mixed"])
@pytest.mark.parametrize("numtask", [2, 7, 12])
def test_task_loss_at_operating_size(golden, fam, numtask):
    """a9 where the reference runs it (slinet.py:167-183 -> loss.py:6-33): stacks of [t+1, 110 592] / [t+1, 73 728] for up to 12 tasks at temperature 0.001 —
    the imported `SliNet.cal_task_loss` on three families of factors (cosines of a few 1e-3: cos / 0.001 decides the loss; cosines near one: saturated,
    the gradient exactly zero; both in one matrix).  Loss and the gradient of the current task's five factors."""
    g = golden("task_loss_wide")
    cfg = synth.VIT_B16
    allf = [{k: torch.from_numpy(v) for k, v in synth.task_family_factors(fam, t, cfg.vision_width, cfg.transformer_width).items()} for t in range(numtask)]
    for v in allf[numtask - 1].values():
        v.requires_grad_(True)
    loss = O.task_loss(numtask - 1, allf, TASK_SIM)
    loss.backward()
    ref = float(g[f"{fam}.{numtask}.loss"])
    assert abs(float(loss.detach()) - ref) <= 2e-5 * max(1.0, abs(ref)), (float(loss.detach()), ref)
    for k in synth.PROMPT_NAMES:
        r = g[f"{fam}.{numtask}.grad.{k}"]
        got = allf[numtask - 1][k].grad.numpy()
        if np.abs(r).max() == 0.0:
            assert np.abs(got).max() == 0.0, k          # saturated: the reference's gradient is exactly zero
        else:
            assert rel_err(got, r) <= 2e-3, (k, rel_err(got, r))


def test_vitb16_last_task_of_a_twelve_task_session(golden):
    """The whole step of task 12 (numtask = 12: base + alignment + task loss over twelve stacks), ViT-B/16, 8 pairs, as the imported reference computes it."""
    cfg = synth.VIT_B16
    orc = O.Oracle(cfg, synth.clip_state_dict(cfg))
    g = golden("vitb16_task12")
    allf = [factors(cfg, t) for t in range(12)]
    res = O.train_step(orc, synth.images(8, 224), g["token_ids"], allf[11], depth=1, numtask=12, all_factors_np=allf, task_sim=TASK_SIM)
    check_step(res, g, tol=5e-5, gtol=2e-3)
    assert abs(res["task_loss"] - g["task_loss"]) <= 2e-5 * max(1.0, abs(g["task_loss"]))


@pytest.mark.parametrize("name,depth", [("vitb16_bs256_d1", 1), ("vitb16_bs256_d3_patched", 3)])
def test_oracle_against_the_256_pair_fixture_on_a_slice_of_the_batch(golden, name, depth):
    """The benchmarked batch (synth.images(256), synth.token_ids(256)) went through the imported reference once (gen_golden.py --only bs256).  A sample's
    features do not depend on its batch (broadcast prompts, slinet.py:116-128), so the oracle on samples [96, 104) must reproduce the fixture's rows
    96..103, and the logits block between them; the full 256-pair step (35 GB of autograd state) is what the GPU tests hold the HIP path to."""
    g = golden(name)
    cfg = synth.VIT_B16
    ids = synth.token_ids(256)
    import zlib
    assert zlib.crc32(np.ascontiguousarray(ids).tobytes()) == int(g["token_ids_crc32"])
    orc = O.Oracle(cfg, synth.clip_state_dict(cfg))
    sl = slice(96, 104)
    fac = {k: torch.from_numpy(v) for k, v in factors(cfg).items()}
    with torch.no_grad():
        img_f, txt_f, _, _ = orc.forward(torch.from_numpy(synth.images(256, 224)[sl]), torch.from_numpy(ids[sl]), fac, depth)
    assert np.abs(img_f.numpy() - g["img_f"][sl]).max() <= 2e-5
    assert np.abs(txt_f.numpy() - g["txt_f"][sl]).max() <= 2e-5
    lg = (orc.W["logit_scale"].exp() * img_f @ txt_f.t()).numpy()
    assert np.abs(lg - g["logits"][sl, sl]).max() <= 1e-4
    # the stored losses follow from the stored logits (ClipLoss, loss.py:75-87)
    L = torch.from_numpy(g["logits"]).double()
    assert abs(float(O.clip_loss(L)) - float(g["base_loss"])) <= 2e-5 * float(g["base_loss"])

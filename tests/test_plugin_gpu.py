"""The plugin surface (SliNet / SPrompts / factory) driving the HIP engine on a real MI355X, checked against the fixtures
captured from the reference (same API calls the reference's hot loop makes: net(images, captions) -> cal_loss -> backward)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from lpi_amd import _lib, synth  # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RET = os.path.join(REPO, "lpi_amd", "retrieval")
DEV = torch.device("cuda:0")


def tiny_args(**over):
    args = json.load(open(os.path.join(RET, "configs", "lpi", "coco_lpi.json")))
    args.update(backbonename="tiny", visual_dim=128, textual_dim=128, device=[DEV], compute_dtype="f32", batch_size=4,
                epochs=1, num_workers=0)
    args.update(over)
    return args


def set_factors(net):
    for t in range(len(net.prompts)):
        for k, v in synth.prompt_factors(9, 16, 128, 128, task=t).items():
            getattr(net.prompts[t], k).data = torch.from_numpy(v.copy()).to(DEV)


def run_step(net, numtask, ids):
    net.numtask = numtask
    net.train()
    for name, p in net.named_parameters():
        p.requires_grad_("prompts." + str(numtask - 1) + "." in name)
        p.grad = None
    img = torch.from_numpy(synth.images(4, 32)).to(DEV)
    n0 = _lib.launch_count()
    img_f, txt_f, vp, tp = net(img, torch.from_numpy(ids))
    out = net.cal_loss(img_f, txt_f, vp, tp)
    loss = sum(v for v in out["loss"].values())
    loss.backward()
    torch.cuda.synchronize()
    assert _lib.launch_count() - n0 > 30
    return img_f, txt_f, vp, tp, out["loss"]


@pytest.mark.parametrize("name,numtask", [("tiny_d1", 1), ("tiny_task2", 2)])
def test_slinet_train_step_matches_reference(golden, name, numtask):
    from lpi_amd.retrieval.models.slinet import SliNet
    g = golden(name)
    net = SliNet(tiny_args()).to(DEV)
    set_factors(net)
    img_f, txt_f, vp, tp, losses = run_step(net, numtask, g["token_ids"])
    assert vp.shape == (4, 9, 16, 128) and vp.stride(0) == 0            # stride-0 batch broadcast like slinet.py:119
    assert np.abs(img_f.detach().cpu().numpy() - g["img_f"]).max() < 1e-4
    assert np.abs(txt_f.detach().cpu().numpy() - g["txt_f"]).max() < 1e-4
    assert set(losses) == ({"base_loss", "alignment_loss"} | ({"task_loss"} if numtask != 1 else set()))
    for k in losses:
        assert abs(float(losses[k]) - float(g[k])) < 1e-4 * max(1.0, abs(float(g[k]))), k
    for k in synth.PROMPT_NAMES:
        got = getattr(net.prompts[numtask - 1], k).grad.cpu().numpy()
        ref = g["grad." + k]
        assert np.abs(got - ref).max() <= 1e-3 * np.abs(ref).max() + 1e-5, k


def test_eval_interfaces_and_task_ids(golden):
    from lpi_amd.retrieval.methods.sprompt import SPrompts
    g = golden("tiny_eval")
    m = SPrompts(tiny_args())
    net = m._network.to(DEV)
    set_factors(net)
    net.numtask = 3
    net.eval()
    img = torch.from_numpy(synth.images(6, 32, seed=synth.IMAGE_SEED + 7)).to(DEV)
    ids = torch.from_numpy(g["token_ids"])
    with torch.no_grad():
        assert np.abs(net.extract_vector(img).cpu().numpy() - g["extract_vector"]).max() < 1e-4
        assert np.abs(net.extract_textual_vector(ids).cpu().numpy() - g["extract_textual_vector"]).max() < 1e-4
        vi = net.visual_interface(img, torch.from_numpy(g["sel_v"]))
        ti = net.textual_interface(ids, torch.from_numpy(g["sel_t"]))
    assert np.abs(vi.cpu().numpy() - g["visual_interface"]).max() < 1e-4
    assert np.abs(ti.cpu().numpy() - g["textual_interface"]).max() < 1e-4
    m.all_keys = m.textual_all_keys = [torch.from_numpy(k).to(DEV) for k in g["task_keys"]]
    assert (m.get_visual_task_id(img).cpu().numpy() == g["visual_task_id"]).all()
    assert (m.get_textual_task_id(ids).cpu().numpy() == g["textual_task_id"]).all()


def test_itm_eval_matches_reference(golden):
    from lpi_amd.retrieval.methods.sprompt import SPrompts
    g = golden("tiny_eval")
    m = SPrompts(tiny_args())
    m.cur_id = 2
    s = g["itm_scores"]
    n_img, n_txt = s.shape
    fr = m.itm_eval(s, s.T.copy(), {t: t // 2 for t in range(n_txt)}, {i: [2 * i, 2 * i + 1] for i in range(n_img)},
                    list(g["itm_cat_i"]), torch.tensor(g["itm_cat_t"]))
    assert np.allclose([fr["mscoco"]["i2t"][t] for t in range(3)], g["itm_i2t"])
    assert np.allclose([fr["mscoco"]["t2i"][t] for t in range(3)], g["itm_t2i"])


def test_incremental_train_two_tasks_end_to_end(tmp_path, monkeypatch):
    """trainer -> factory -> SPrompts.incremental_train over synthetic loaders: the whole continual loop incl. task_loss,
    KMeans task keys, retrieval eval and the final_res JSON (sprompt.py:150-187, 638-646)."""
    monkeypatch.chdir(tmp_path)
    from lpi_amd.retrieval import trainer
    args = tiny_args(num_tasks=2, synthetic_train_size=8, synthetic_eval_images_per_task=6, seed=[1993], device=["0"])
    before = None
    model = trainer._train(args)
    net = model._network
    assert net.numtask == 2 and len(model.all_keys) == 2 and model.all_keys[0].shape == (5, 128)
    fr = model.final_res
    assert set(fr) == {0, 1} and set(fr[1]["mscoco"]) == {"i2t", "t2i"} and set(fr[1]["mscoco"]["i2t"]) == {0, 1}
    assert all(len(v) == 3 and 0 <= v[0] <= v[1] <= v[2] <= 100 for v in fr[1]["mscoco"]["t2i"].values())
    assert len(list((tmp_path / "res").glob("*.json"))) == 1
    ref0 = synth.prompt_factors(9, 16, 128, 128, task=0)["dim_1_share"]
    assert model._old_network is not None and model._old_network.engine is net.engine
    # task-1 prompts moved (trained), task-5 prompts did not
    p5 = net.prompts[5].dim_1_share.detach().clone()
    assert net.prompts[1].dim_1_share.grad is not None and net.prompts[5].dim_1_share.grad is None


def test_empty_batch_fails_loudly():
    """An empty batch is an error, not a silent no-op (there is no CPU fallback to absorb it)."""
    from lpi_amd import _lib, synth
    from lpi_amd.engine import DualEncoder
    cfg = synth.TINY
    enc = DualEncoder(cfg, synth.clip_state_dict(cfg), dtype="f32", device="cuda:0")
    with pytest.raises((_lib.LpiError, ValueError, RuntimeError)):
        enc.encode_image(torch.zeros(0, 3, cfg.image_resolution, cfg.image_resolution, device="cuda:0"))
    with pytest.raises((_lib.LpiError, ValueError, RuntimeError)):
        enc.encode_text(torch.zeros(0, 77, dtype=torch.int64, device="cuda:0"))


def test_parameters_after_three_sgd_cosine_steps_match_reference(golden, monkeypatch):
    """a10: not only the gradients but the PARAMETERS after k optimiser steps — SGD(momentum .9, lr .05, wd 2e-4) over
    network.parameters() with CosineAnnealingLR(T_max = epochs) stepped per epoch (sprompt.py:253-255, 311, 324) — through the plugin's
    own _train / train_function, three epochs of one batch, against the reference's parameters after each... the last step."""
    from lpi_amd.retrieval.methods.sprompt import SPrompts
    g = golden("tiny_sgd3")
    steps = int(g["steps"])
    m = SPrompts(tiny_args(epochs=steps, lrate=float(g["lrate"]), weight_decay=float(g["weight_decay"])))
    net = m._network.to(DEV)
    set_factors(net)
    net.numtask = 1
    ids = torch.from_numpy(golden("tiny_d1")["token_ids"])          # same four captions as the fixture's batch
    img = torch.from_numpy(synth.images(4, 32))
    monkeypatch.setattr(m, "clustering", lambda dataloader: None)
    monkeypatch.setattr(m, "_evaluate_retrieval", lambda loader: (None, None, {}))
    m._train([(img, ids, 0, 0)], None)
    for k in synth.PROMPT_NAMES:
        got = getattr(net.prompts[0], k).detach().cpu().numpy()
        ref = g[f"param.{steps - 1}.{k}"]
        start = synth.prompt_factors(9, 16, 128, 128, task=0)[k]
        moved = np.abs(ref - start).max()
        assert moved > 0                                             # the fixture's parameters did move
        assert np.abs(got - ref).max() <= 2e-3 * moved + 1e-6, (k, np.abs(got - ref).max(), moved)


def test_nt_bxent_loss_has_no_cpu_fallback():
    from lpi_amd.retrieval.loss.loss import nt_bxent_loss
    with pytest.raises(_lib.LpiError):
        nt_bxent_loss(torch.randn(3, 8), torch.eye(3), 0.001)


def test_flat_sgd_equals_torch_sgd_and_flat_gradients_are_the_autograd_gradients():
    """optim.FlatSGD (one lpi_sgd_step launch over the five factors laid out in one flat vector) against torch.optim.SGD with the reference's
    hyper-parameters and cosine schedule (sprompt.py:253-255) over four steps of the tiny model; the gradients that train_step writes into the
    flat vector (functional.DecomposedPromptFn grad_out) are the .grad tensors themselves and equal the gradients of the loss-graph path
    (forward_loss + backward), bit for bit."""
    from lpi_amd.engine import DualEncoder
    from lpi_amd.optim import CosineLR, FlatSGD, flatten
    from lpi_amd.step import forward_loss, train_step
    cfg = synth.TINY
    enc = DualEncoder(cfg, synth.clip_state_dict(cfg), dtype="f32", device=DEV)
    img = torch.from_numpy(synth.images(4, cfg.image_resolution)).to(DEV)
    ids = torch.from_numpy(synth.token_ids(4)).to(DEV)
    fac_np = synth.prompt_factors(9, 16, cfg.vision_width, cfg.transformer_width)
    fa = {k: torch.from_numpy(v).to(DEV).requires_grad_(True) for k, v in fac_np.items()}
    fb = {k: torch.from_numpy(v).to(DEV).requires_grad_(True) for k, v in fac_np.items()}
    flat, flat_grad, views = flatten(fa)
    oa = FlatSGD(fa, lr=0.05, momentum=0.9, weight_decay=2e-4, flat=flat, flat_grad=flat_grad, grad_views=views)
    sa = CosineLR(oa, T_max=4)
    ob = torch.optim.SGD(list(fb.values()), lr=0.05, momentum=0.9, weight_decay=2e-4)
    sb = torch.optim.lr_scheduler.CosineAnnealingLR(ob, T_max=4)
    for step in range(4):
        out = train_step(enc, img, ids, fa, 2, flat_grad=flat_grad, grad_views=views)
        for k, v in zip(fa, views):
            assert fa[k].grad.data_ptr() == v.data_ptr()              # the .grad tensors ARE the slices of the flat gradient
        ob.zero_grad()
        losses, *_ = forward_loss(enc, img, ids, fb, 2)
        (losses["base_loss"] + losses["alignment_loss"]).backward()
        assert abs(float(out["base_loss"]) - float(losses["base_loss"])) < 1e-6 and abs(float(out["alignment_loss"]) - float(losses["alignment_loss"])) < 1e-6
        for k in fa:     # bit for bit while the parameters are the same bits (the two optimisers round differently in the last place)
            if step == 0:
                assert torch.equal(fa[k].grad, fb[k].grad), (step, k)
            assert float((fa[k].grad - fb[k].grad).abs().max()) <= 1e-4 * float(fb[k].grad.abs().max()) + 1e-9, (step, k)
        oa.step(); sa.step()
        ob.step(); sb.step()
        assert abs(oa.param_groups[0]["lr"] - ob.param_groups[0]["lr"]) < 1e-9
        for k in fa:
            assert float((fa[k] - fb[k]).abs().max()) <= 1e-6 * float(fb[k].abs().max()), (step, k)


def test_interact_module_vs_reference_fixture_and_oracle(golden):
    """SURVEY 8 (f4, optional): lpi_amd's InteractModule (lpi_interact_fwd / _bwd, rank-r form) with the reference module's parameters against the
    outputs and autograd gradients of the imported reference class (tests/golden/interact.npz) — outputs 1e-5, every gradient (six factors, both
    LayerNorms' affine, both inputs) 1e-4 relative — and the reference's parameter names / shapes / initialisation contract."""
    from lpi_amd.retrieval.models.prompts.prompts import InteractModule
    g = golden("interact")
    bs, P, Dv, Dt, layer_num, r = (int(x) for x in g["shape"])
    inp = synth.interact_inputs(bs, P, Dv, Dt)
    m = InteractModule(layer_num=layer_num, visual_dim=Dv, textual_dim=Dt, r=r)
    names = {n: tuple(p.shape) for n, p in m.named_parameters()}
    assert names == {k[6:]: tuple(g[k].shape) for k in g if k.startswith("param.")}          # the reference's names and shapes
    bound = 1 / np.sqrt(r)                                                                      # kaiming_uniform_(a = sqrt 5): U(-1/sqrt(fan_in), ..)
    assert float(m.dim_2_v2t.detach().abs().max()) <= bound + 1e-6 and float(m.dim_3_t2v.detach().abs().max()) <= bound + 1e-6
    with torch.no_grad():
        for n, p in m.named_parameters():
            p.copy_(torch.from_numpy(g["param." + n]))
    m = m.to(DEV)
    v = torch.from_numpy(inp["visual_in"]).to(DEV).requires_grad_(True)
    t = torch.from_numpy(inp["textual_in"]).to(DEV).requires_grad_(True)
    n0 = _lib.launch_count()
    vo, to = m(v, t, int(g["layer_id"]))
    ((vo * torch.from_numpy(inp["wv"]).to(DEV)).sum() + (to * torch.from_numpy(inp["wt"]).to(DEV)).sum()).backward()
    torch.cuda.synchronize()
    assert _lib.launch_count() - n0 >= 3
    assert np.abs(vo.detach().cpu().numpy() - g["visual_out"]).max() < 1e-5
    assert np.abs(to.detach().cpu().numpy() - g["textual_out"]).max() < 1e-5
    grads = {"visual_in": v.grad, "textual_in": t.grad, **{n: p.grad for n, p in m.named_parameters()}}
    for n, got in grads.items():
        ref = g["grad." + n]
        e = np.abs(got.cpu().numpy() - ref).max()
        assert e <= 1e-4 * np.abs(ref).max() + 1e-7, (n, e, np.abs(ref).max())


def relerr(got, ref):
    got, ref = got.double().cpu(), ref.double().cpu()
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-30))


@pytest.mark.parametrize("N,Dv,Dt,r,layer_num,layer", [(1, 64, 128, 1, 2, 0), (37, 96, 768, 4, 9, 8), (300, 1024, 768, 8, 12, 5), (513, 768, 512, 3, 3, 1)])
def test_interact_kernels_vs_f64(N, Dv, Dt, r, layer_num, layer):
    """lpi_interact_fwd / _bwd against the oracle's restatement in f64 with autograd, at widths up to the kernels' limit (1024), ranks 1 .. 8, row
    counts that leave partial workgroups; the backward twice: bitwise the same (fixed summation order, no atomics)."""
    from lpi_amd.functional import InteractFn
    from oracle import lpi_oracle as O
    gen = torch.Generator().manual_seed(N * 7 + r)
    rn = lambda *s: torch.randn(*s, generator=gen)  # noqa: E731
    p = {"dim_1_v2t": rn(layer_num, r), "dim_2_v2t": rn(Dv + 1, r) * 0.2, "dim_3_v2t": rn(Dt, r), "dim_1_t2v": rn(layer_num, r),
         "dim_2_t2v": rn(Dt + 1, r) * 0.2, "dim_3_t2v": rn(Dv, r), "visual_norm.weight": 1 + 0.1 * rn(Dv), "visual_norm.bias": 0.1 * rn(Dv),
         "textual_norm.weight": 1 + 0.1 * rn(Dt), "textual_norm.bias": 0.1 * rn(Dt)}
    xv, xt, wv, wt = rn(N, Dv), rn(N, Dt), rn(N, Dv), rn(N, Dt)
    order = ("dim_1_v2t", "dim_2_v2t", "dim_3_v2t", "dim_1_t2v", "dim_2_t2v", "dim_3_t2v", "visual_norm.weight", "visual_norm.bias",
             "textual_norm.weight", "textual_norm.bias")
    runs = []
    for _ in range(2):
        pd = {k: v.clone().to(DEV).requires_grad_(True) for k, v in p.items()}
        a, b = xv.clone().to(DEV).requires_grad_(True), xt.clone().to(DEV).requires_grad_(True)
        vo, to = InteractFn.apply(a, b, layer, *[pd[k] for k in order])
        ((vo * wv.to(DEV)).sum() + (to * wt.to(DEV)).sum()).backward()
        runs.append({"vo": vo.detach(), "to": to.detach(), "xv": a.grad, "xt": b.grad, **{k: pd[k].grad for k in order}})
    for k in runs[0]:
        assert torch.equal(runs[0][k], runs[1][k]), k
    p64 = {k: v.double().requires_grad_(True) for k, v in p.items()}
    a64, b64 = xv.double().requires_grad_(True), xt.double().requires_grad_(True)
    vr, tr = O.interact(p64, a64, b64, layer)
    ((vr * wv.double()).sum() + (tr * wt.double()).sum()).backward()
    assert relerr(runs[0]["vo"], vr.detach()) < 2e-5 and relerr(runs[0]["to"], tr.detach()) < 2e-5
    assert relerr(runs[0]["xv"], a64.grad) < 1e-4 and relerr(runs[0]["xt"], b64.grad) < 1e-4
    for k in order:
        assert relerr(runs[0][k], p64[k].grad) < 1e-4, k


def test_engine_and_its_workspace_are_freed_after_a_training_step():
    """The autograd nodes of the towers keep the engine and its workspace arena (tens of GB at full size) in their context; a tensor that is both a
    node's output and held by that node's context would be a reference cycle through a C++ object that Python's collector cannot break.  After a
    step, dropping the engine must return its memory."""
    import gc
    from lpi_amd.engine import DualEncoder, PackedIds
    from lpi_amd.step import train_step
    cfg = synth.TINY
    # what a process keeps after its FIRST engine is not the engine's: the vendor BLAS handle's workspace (the weight folding at construction multiplies
    # in f64 on the device: ~128 MB on first use) and the module-level scratch (split-K partials, loss workspace).  Build and run one engine first, so
    # that this test measures the same thing whether it runs alone or behind the other files
    warm = DualEncoder(cfg, synth.clip_state_dict(cfg), dtype="bf16", device=DEV)
    wf = {k: torch.from_numpy(v).to(DEV).requires_grad_(True) for k, v in synth.prompt_factors(9, 16, cfg.vision_width, cfg.transformer_width).items()}
    train_step(warm, torch.from_numpy(synth.images(64, cfg.image_resolution)).to(DEV), PackedIds(synth.token_ids(64)).to(DEV), wf, 2)
    torch.cuda.synchronize()
    del warm, wf
    gc.collect()
    torch.cuda.empty_cache()
    base = torch.cuda.memory_allocated()
    enc = DualEncoder(cfg, synth.clip_state_dict(cfg), dtype="bf16", device=DEV)
    fac = {k: torch.from_numpy(v).to(DEV).requires_grad_(True) for k, v in synth.prompt_factors(9, 16, cfg.vision_width, cfg.transformer_width).items()}
    img = torch.from_numpy(synth.images(64, cfg.image_resolution)).to(DEV)
    ids = PackedIds(synth.token_ids(64)).to(DEV)
    out = train_step(enc, img, ids, fac, 2)
    torch.cuda.synchronize()
    used = torch.cuda.memory_allocated() - base
    assert used > (8 << 20)
    del enc, fac, img, ids, out
    gc.collect()
    assert not [o for o in gc.get_objects() if type(o).__name__ == "DualEncoder"]
    assert torch.cuda.memory_allocated() - base < max(used // 8, 8 << 20)          # the caches existed before `base`: (almost) everything comes back


def test_slinet_takes_caption_strings_through_the_native_tokenizer(tmp_path, monkeypatch):
    """SliNet.forward(image, list[str]) on a box WITHOUT CLIP's merge table (the GPU box): a seeded synthetic table of the same format
    (tests/bpe_synth.py) stands in through $LPI_BPE_VOCAB.  The C++ tokenizer (lpi_bpe_*) and the Python one give the same ids for the
    PromptLearner's "X X ... caption." strings (prompt_learner.py:131, clip.py:185-221), and the plugin's features from the strings equal its
    features from those ids."""
    import bpe_synth
    from lpi_amd.retrieval.models.clip import prompt_learner as PL
    from lpi_amd.retrieval.models.clip import simple_tokenizer as T
    from lpi_amd.retrieval.models.slinet import SliNet
    path = bpe_synth.write_table(tmp_path / "synthetic_bpe.txt.gz", seed=5)
    monkeypatch.setenv("LPI_BPE_VOCAB", path)
    monkeypatch.setattr(PL, "_tokenizer", None)
    caps = ["a photo of a dog", "two people riding bicycles down the street", "it's a naïve café, isn't it?", "3 zebras & 2 giraffes 😀"]
    net = SliNet(tiny_args()).to(DEV)
    set_factors(net)
    net.numtask = 1
    net.eval()
    img = torch.from_numpy(synth.images(4, 32)).to(DEV)
    n0 = _lib.launch_count()
    with torch.no_grad():
        img_s, txt_s, _, _ = net(img, caps)
    assert isinstance(PL.get_tokenizer(), T.NativeTokenizer) and _lib.launch_count() > n0
    prompts = [" ".join(["X"] * 16) + " " + c + "." for c in caps]
    ids_native = T.tokenize(PL.get_tokenizer(), prompts)
    ids_py = T.tokenize(T.SimpleTokenizer(path), prompts)
    assert (ids_native.numpy() == ids_py.numpy()).all()
    x = int(ids_native[0, 1])
    assert (ids_native[:, 0] == 49406).all() and (ids_native[:, 1:17] == x).all() and (ids_native.max(1).values == 49407).all()
    with torch.no_grad():
        img_i, txt_i, _, _ = net(img, ids_py)
    assert torch.equal(txt_s, txt_i) and torch.equal(img_s, img_i)
    monkeypatch.setattr(PL, "_tokenizer", None)      # the next user of the module looks its table up again

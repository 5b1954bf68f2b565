"""Parity at BASELINE.json's full size (ViT-B/16, B = 256/GPU, depth 3) through size-independent properties — the oracle is too slow
at this size, so the HIP path is checked against invariants instead: unit-norm features, batch-composition invariance, the loss
recomputed on the CPU from the features, finite differences of the loss w.r.t. prompt factors, and data-parallel equivalence
(virtual ranks on one GPU: all-gather + local-rows gradient + SUM == single-rank global batch)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from lpi_amd import _lib, synth  # noqa: E402
from lpi_amd.engine import DualEncoder  # noqa: E402
from lpi_amd.step import forward_loss, train_step  # noqa: E402

DEV = "cuda:0"
CFG = synth.VIT_B16
B = 256


@pytest.fixture(scope="module")
def enc32():
    return DualEncoder(CFG, synth.clip_state_dict(CFG), dtype="f32", device=DEV)


def factors(requires_grad=True, r=4):
    return {k: torch.from_numpy(v).to(DEV).requires_grad_(requires_grad)
            for k, v in synth.prompt_factors(9, 16, CFG.vision_width, CFG.transformer_width, r=r).items()}


@pytest.fixture(scope="module")
def data():
    return (torch.from_numpy(synth.images(B, 224)).to(DEV), torch.from_numpy(synth.token_ids(B)).to(DEV))


def test_full_batch_step_properties(enc32, data):
    img, ids = data
    fac = factors()
    out = train_step(enc32, img, ids, fac, 3)
    i_f, t_f = out["img_f"].double().cpu(), out["txt_f"].double().cpu()
    assert torch.allclose(i_f.norm(dim=1), torch.ones(B, dtype=torch.float64), atol=1e-5)
    assert torch.allclose(t_f.norm(dim=1), torch.ones(B, dtype=torch.float64), atol=1e-5)
    lg = enc32.logit_scale_exp * i_f @ t_f.t()
    lab = torch.arange(B)
    ce = (torch.nn.functional.cross_entropy(lg, lab) + torch.nn.functional.cross_entropy(lg.t(), lab)) / 2
    assert abs(float(out["base_loss"]) - float(ce)) < 2e-5 * max(1.0, float(ce))
    for k in synth.PROMPT_NAMES:
        g = fac[k].grad
        assert g is not None and torch.isfinite(g).all() and float(g.abs().max()) > 0


def test_batch_composition_invariance(enc32, data):
    """A sample's features do not depend on what else is in the batch (row padding, tile boundaries, head/tile scheduling)."""
    img, ids = data
    fac = factors(False)
    with torch.no_grad():
        _, f256, t256, _, _ = forward_loss(enc32, img, ids, fac, 3)
        f256, t256 = f256.clone(), t256.clone()
        _, f9, t9, _, _ = forward_loss(enc32, img[100:109], ids[100:109], fac, 3)
    assert float((f256[100:109] - f9).abs().max()) < 1e-6
    assert float((t256[100:109] - t9).abs().max()) < 1e-6


def test_finite_difference_of_total_loss(enc32, data):
    """d(base + alignment loss)/d(factor entry) from the hand-written backward vs central differences, in f32 at full size."""
    img, ids = data
    fac = factors()
    train_step(enc32, img, ids, fac, 3)
    grads = {k: fac[k].grad.clone() for k in synth.PROMPT_NAMES}

    def total(f):
        with torch.no_grad():
            losses, *_ = forward_loss(enc32, img, ids, f, 3)
        return float(losses["base_loss"].double() + losses["alignment_loss"].double())

    for name, idx in (("dim_1_share", (0, 1)), ("dim_2_visual", (3, 2)), ("dim_3_textual", (17, 0)), ("dim_1_share", (2, 3))):
        eps = 2e-2
        f = factors(False)
        f[name][idx] += eps
        lp = total(f)
        f[name][idx] -= 2 * eps
        lm = total(f)
        fd = (lp - lm) / (2 * eps)
        g = float(grads[name][idx])
        assert abs(fd - g) <= 0.05 * abs(g) + 2e-4, (name, idx, fd, g)


class _VirtualExchange:
    """Stand-in for dp.Exchange on ONE GPU: rank r of W sees the (pre-computed) features of all ranks."""

    def __init__(self, rank, world, all_img, all_txt, per):
        self.rank, self.world, self.all_img, self.all_txt, self.per = rank, world, all_img, all_txt, per

    def gather(self, img_f, txt_f):
        return self.all_img.clone(), self.all_txt.clone(), self.rank * self.per

    def allreduce_grads(self, params):
        return 0


def test_data_parallel_equivalence_on_virtual_ranks(enc32, data):
    """W = 4 virtual ranks of 64 pairs: sum of per-rank factor gradients (global loss, gradient through local rows only, alignment
    term scaled by 1/W) equals the single-rank gradient on the 256-pair batch — the DP math on the real HIP path."""
    img, ids = data
    W, per = 4, B // 4
    fac = factors()
    ref = train_step(enc32, img, ids, fac, 3)
    gref = {k: fac[k].grad.clone() for k in synth.PROMPT_NAMES}
    all_img, all_txt = ref["img_f"].clone(), ref["txt_f"].clone()
    acc = {k: torch.zeros_like(v) for k, v in gref.items()}
    for r in range(W):
        f = factors()
        ex = _VirtualExchange(r, W, all_img, all_txt, per)
        out = train_step(enc32, img[r * per:(r + 1) * per], ids[r * per:(r + 1) * per], f, 3, exchange=ex)
        assert abs(float(out["base_loss"]) - float(ref["base_loss"])) < 1e-5
        for k in synth.PROMPT_NAMES:
            acc[k] += f[k].grad
    for k in synth.PROMPT_NAMES:
        scale = float(gref[k].abs().max())
        assert float((acc[k] - gref[k]).abs().max()) <= 2e-4 * scale + 1e-7, k


def test_bf16_full_size_close_to_f32(enc32, data):
    img, ids = data
    fac = factors(False)
    with torch.no_grad():
        _, f32i, f32t, _, _ = forward_loss(enc32, img, ids, fac, 3)
        f32i, f32t = f32i.clone(), f32t.clone()
    encb = DualEncoder(CFG, synth.clip_state_dict(CFG), dtype="bf16", device=DEV)
    with torch.no_grad():
        _, bi, bt, _, _ = forward_loss(encb, img, ids, fac, 3)
    assert float((bi - f32i).abs().max()) < 5e-3 and float((bt - f32t).abs().max()) < 5e-3
    # retrieval decisions: top-1 agrees wherever the f32 margin is not tiny
    s32, sb = (f32i @ f32t.t()), (bi @ bt.t())
    top2 = s32.topk(2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > 2e-3
    assert safe.float().mean() > 0.5
    assert (s32.argmax(1)[safe] == sb.argmax(1)[safe]).all()


def test_vit_l14_depth12_rank8_vs_oracle():
    """BASELINE configs[4]'s architecture in full (ViT-L/14: 24 vision layers of width 1024 / 16 heads, 12 text layers of width 768,
    patch 14 -> 257 + 16 tokens, embed 768; prompt_depth 12, CP rank 8) at batch 2, f32 mode, against the f32 oracle: features and
    losses to 1e-4, factor gradients to 5e-3 relative (ATen-vs-MFMA summation order over 24 layers)."""
    import numpy as np
    from lpi_amd import synth
    from lpi_amd.engine import DualEncoder, trim_token_ids
    from lpi_amd.step import train_step
    from oracle import lpi_oracle as O
    cfg = synth.VIT_L14
    sd = synth.clip_state_dict(cfg)
    fac_np = synth.prompt_factors(12, 16, cfg.vision_width, cfg.transformer_width, r=8)
    img, ids = synth.images(2, cfg.image_resolution), synth.token_ids(2)
    ref = O.train_step(O.Oracle(cfg, sd, torch.float32), img, ids, fac_np, depth=12)
    enc = DualEncoder(cfg, sd, dtype="f32", device="cuda:0")
    fac = {k: torch.from_numpy(v).to("cuda:0").requires_grad_(True) for k, v in fac_np.items()}
    out = train_step(enc, torch.from_numpy(img).to("cuda:0"), torch.from_numpy(np.ascontiguousarray(trim_token_ids(ids))).to("cuda:0"), fac, 12)
    for k in ("img_f", "txt_f", "base_loss", "alignment_loss"):
        err = float(np.abs(out[k].cpu().numpy() - ref[k]).max())
        assert err <= 1e-4, (k, err)
    for k in synth.PROMPT_NAMES:
        g, r = fac[k].grad.cpu().numpy(), ref["grad." + k]
        err = float(np.abs(g - r).max())
        assert err <= 5e-3 * np.abs(r).max() + 1e-7, (k, err, float(np.abs(r).max()))


def test_vit_l14_bf16_mode_close_to_oracle():
    """BASELINE configs[4] names ViT-L/14, depth 12, r = 8 in BF16: the throughput mode on that architecture (L = 273: the two-pass
    attention backward, 1024 / 768-wide GEMMs, 16 / 12 heads) at batch 3 against the f32 oracle, with the bf16-mode bars."""
    import numpy as np
    from oracle import lpi_oracle as O
    cfg = synth.VIT_L14
    sd = synth.clip_state_dict(cfg)
    fac_np = synth.prompt_factors(12, 16, cfg.vision_width, cfg.transformer_width, r=8)
    img, ids = synth.images(3, cfg.image_resolution), synth.token_ids(3)
    ref = O.train_step(O.Oracle(cfg, sd, torch.float32), img, ids, fac_np, depth=12)
    enc = DualEncoder(cfg, sd, dtype="bf16", device=DEV)
    fac = {k: torch.from_numpy(v).to(DEV).requires_grad_(True) for k, v in fac_np.items()}
    out = train_step(enc, torch.from_numpy(img).to(DEV), torch.from_numpy(ids).to(DEV), fac, 12)
    for k in ("img_f", "txt_f"):
        err = float(np.abs(out[k].cpu().numpy() - ref[k]).max())
        assert err <= 2e-2, (k, err)
    assert abs(float(out["base_loss"]) - float(ref["base_loss"])) <= 3e-2 * max(1.0, abs(float(ref["base_loss"])))
    assert abs(float(out["alignment_loss"]) - float(ref["alignment_loss"])) <= 1e-4
    cos = lambda a, b: float((a * b).sum() / np.sqrt((a * a).sum() * (b * b).sum()))  # noqa: E731
    for k in synth.PROMPT_NAMES:
        c = cos(fac[k].grad.double().cpu().numpy(), ref["grad." + k].astype(np.float64))
        rel = float(np.abs(fac[k].grad.double().cpu().numpy() - ref["grad." + k]).max() / np.abs(ref["grad." + k]).max())
        print("ViT-L/14 bf16 vs the f32 oracle", k, "cosine", round(c, 5), "max relative error", round(rel, 4))
        assert c >= 0.998 and rel <= 8e-2, (k, c, rel)       # measured >= 0.99900 / <= 4.7e-2
    del enc
    torch.cuda.empty_cache()


def test_vit_l14_336px_vs_oracle():
    """The last CLIP ViT of clip.available_models() (clip.py:30-40): ViT-L/14@336px — 577 vision tokens + 16 prompts, more than the one-workgroup-per-(sample,
    head) attention kernels hold; every block but the last runs on csrc/attn_long.hip (tiled over the keys, online softmax), the last on the pooled-row path.
    Batch 2, depth 3, r 8: f32 mode against the f32 oracle at the ViT-L/14 bars, bf16 mode at its bars."""
    import numpy as np
    from oracle import lpi_oracle as O
    cfg = synth.VIT_L14_336
    sd = synth.clip_state_dict(cfg)
    fac_np = synth.prompt_factors(12, 16, cfg.vision_width, cfg.transformer_width, r=8)
    img, ids = synth.images(2, cfg.image_resolution), synth.token_ids(2)
    ref = O.train_step(O.Oracle(cfg, sd, torch.float32), img, ids, fac_np, depth=3)
    cos = lambda a, b: float((a * b).sum() / np.sqrt((a * a).sum() * (b * b).sum()))  # noqa: E731
    for mode in ("f32", "bf16"):
        enc = DualEncoder(cfg, sd, dtype=mode, device=DEV)
        fac = {k: torch.from_numpy(v).to(DEV).requires_grad_(True) for k, v in fac_np.items()}
        n0 = _lib.launch_count()
        out = train_step(enc, torch.from_numpy(img).to(DEV), torch.from_numpy(ids).to(DEV), fac, 3)
        torch.cuda.synchronize()
        assert _lib.launch_count() > n0
        ftol = 1e-4 if mode == "f32" else 2e-2
        for k in ("img_f", "txt_f"):
            err = float(np.abs(out[k].cpu().numpy() - ref[k]).max())
            assert err <= ftol, (mode, k, err)
        assert abs(float(out["base_loss"]) - float(ref["base_loss"])) <= (1e-4 if mode == "f32" else 3e-2) * max(1.0, abs(float(ref["base_loss"])))
        for k in synth.PROMPT_NAMES:
            g, r = fac[k].grad.double().cpu().numpy(), ref["grad." + k].astype(np.float64)
            rel = float(np.abs(g - r).max() / np.abs(r).max())
            print(f"ViT-L/14@336px {mode} vs the f32 oracle", k, "cosine", round(cos(g, r), 5), "max relative error", round(rel, 5))
            if mode == "f32":
                assert rel <= 5e-3, (k, rel)
            else:
                # two pairs only, 24 layers: dim_1_share measured 0.9984 / 9.4e-2; the long-sequence kernels themselves err like the short ones
                # (tools/probe/attn_long_err.py: rms 2.4e-3 of dq / dk / dv at L = 273 and at L = 586)
                assert cos(g, r) >= 0.997 and rel <= 0.15, (k, cos(g, r), rel)
        del enc
        torch.cuda.empty_cache()


def test_eight_rank_global_loss_at_configs3_size(enc32, data):
    """BASELINE configs[3]: 8 ranks x 256 pairs -> the 2048 x 2048 contrastive matrix.  Eight virtual ranks' f32 HIP features (8
    forward passes of 256 pairs with per-rank seeds, as bench.py draws them) are gathered into the [2048, 1024] buffer the RCCL
    all-gather produces; every rank's loss kernels (full global loss, gradient of its 256 local rows only) are checked against f64
    autograd on the global batch, and the ranks' row blocks tile the global gradient exactly once."""
    W = 8
    fac = factors(False)
    buf = torch.empty(W * B, 2 * CFG.embed_dim, device=DEV)
    with torch.no_grad():
        for r in range(W):
            im = torch.from_numpy(synth.images(B, 224, seed=synth.IMAGE_SEED + r)).to(DEV)
            tk = torch.from_numpy(synth.token_ids(B, seed=synth.TOKEN_SEED + r)).to(DEV)
            _, fi, ft, _, _ = forward_loss(enc32, im, tk, fac, 3)
            buf[r * B:(r + 1) * B, :CFG.embed_dim] = fi
            buf[r * B:(r + 1) * B, CFG.embed_dim:] = ft
    from lpi_amd.engine import clip_loss_fwd_bwd
    i64 = buf[:, :CFG.embed_dim].double().cpu().requires_grad_(True)
    t64 = buf[:, CFG.embed_dim:].double().cpu().requires_grad_(True)
    lg = enc32.logit_scale_exp * i64 @ t64.t()
    lab = torch.arange(W * B)
    ref = (torch.nn.functional.cross_entropy(lg, lab) + torch.nn.functional.cross_entropy(lg.t(), lab)) / 2
    ref.backward()
    for r in range(W):
        loss, logits, dI, dT = clip_loss_fwd_bwd(buf[:, :CFG.embed_dim], buf[:, CFG.embed_dim:], enc32.logit_scale_exp, True, r * B, B)
        assert logits.shape == (W * B, W * B)
        assert abs(float(loss) - float(ref)) < 2e-5 * max(1.0, float(ref))
        assert float((dI.double().cpu() - i64.grad[r * B:(r + 1) * B]).abs().max()) < 1e-7 + 1e-4 * float(i64.grad.abs().max())
        assert float((dT.double().cpu() - t64.grad[r * B:(r + 1) * B]).abs().max()) < 1e-7 + 1e-4 * float(t64.grad.abs().max())


@pytest.mark.parametrize("mode,ftol,ltol_rel", [("bf16", 5e-3, 2e-2), ("f16", 1.5e-3, 5e-3)])
def test_throughput_modes_train_step_at_the_benchmarked_configuration(enc32, data, mode, ftol, ltol_rel):
    """The exact configuration bench.py times — bf16 operands, B = 256, depth 3, text batch trimmed to the longest caption (so the
    256x256 bf16 kernel with its hybrid tail, the 256x128 kernel, the fused attention backward at L = 213 / 59 and the fp16 residual
    stream are all on the path) — against the f32 HIP step on the same inputs (which the fixtures pin to the reference)."""
    from lpi_amd.engine import trim_token_ids
    img, ids = data
    ids_t = torch.from_numpy(np.ascontiguousarray(trim_token_ids(synth.token_ids(B)))).to(DEV)
    assert ids_t.shape[1] < ids.shape[1]
    f32, fb = factors(), factors()
    o32 = train_step(enc32, img, ids, f32, 3)                      # untrimmed f32 = the parity path
    o32 = {k: v.clone() for k, v in o32.items()}
    encb = DualEncoder(CFG, synth.clip_state_dict(CFG), dtype=mode, device=DEV)      # "f16": fp16 operands forward, bf16 gradient stream
    ob = train_step(encb, img, ids_t, fb, 3)
    torch.cuda.synchronize()
    cos = lambda a, b: float((a * b).sum() / (a.norm() * b.norm()))  # noqa: E731
    report = {}
    for k in synth.PROMPT_NAMES:
        a, b = fb[k].grad.double().cpu(), f32[k].grad.double().cpu()
        report[k] = (cos(a, b), float((a - b).abs().max() / b.abs().max()))
        assert report[k][0] >= 0.9995, (k, report[k])        # measured >= 0.99994
        assert report[k][1] <= 0.03, (k, report[k])          # measured <= 1.5e-2 (bf16), <= 1.1e-2 (f16) at B = 256, depth 3
    print(f"{mode} vs f32 factor gradients (cosine, max rel err):", {k: (round(c, 5), round(r, 4)) for k, (c, r) in report.items()})
    for k, tol in (("base_loss", ltol_rel), ("alignment_loss", 1e-5)):
        assert abs(float(ob[k]) - float(o32[k])) <= tol * max(1.0, abs(float(o32[k]))), (k, float(ob[k]), float(o32[k]))
    assert float((ob["img_f"] - o32["img_f"]).abs().max()) < ftol and float((ob["txt_f"] - o32["txt_f"]).abs().max()) < ftol
    # top-1 retrieval agreement wherever the f32 margin exceeds the measured bf16 logit error
    l32 = (enc32.logit_scale_exp * o32["img_f"] @ o32["txt_f"].t()).cpu()
    lb = (encb.logit_scale_exp * ob["img_f"] @ ob["txt_f"].t()).cpu()
    err = float((lb - l32).abs().max())
    top2 = l32.topk(2, dim=1).values
    safe = (top2[:, 0] - top2[:, 1]) > 2 * err
    assert torch.equal(lb.argmax(1)[safe], l32.argmax(1)[safe])
    print(f"{mode} logits: max |err| {err:.3e}; top-1 asserted on {int(safe.sum())} of {B} rows")
    del encb
    torch.cuda.empty_cache()


@pytest.mark.parametrize("mode", ["bf16", "f16"])
def test_lockstep_towers_with_grouped_gemm_launches_equal_the_sequential_towers(data, mode):
    """The default step runs the two towers in lock step and issues their GEMMs of the same layer op as ONE grouped persistent launch
    (lpi_gemm_nt_grouped; engine.run_lockstep).  Same kernels on the same operands: features, losses and factor gradients are the same
    BITS as with the towers one after the other (slinet.py:121-133's order), and the grouped path really ran (fewer launches)."""
    from lpi_amd import _lib
    from lpi_amd.engine import trim_token_ids
    img, _ = data
    ids_t = torch.from_numpy(np.ascontiguousarray(trim_token_ids(synth.token_ids(B)))).to(DEV)
    enc = DualEncoder(CFG, synth.clip_state_dict(CFG), dtype=mode, device=DEV)
    res = {}
    for lock in (True, False):
        fac = factors()
        n0 = _lib.launch_count()
        out = train_step(enc, img, ids_t, fac, 3, lockstep=lock)
        torch.cuda.synchronize()
        res[lock] = ({k: v.clone() for k, v in out.items()}, {k: fac[k].grad.clone() for k in synth.PROMPT_NAMES}, _lib.launch_count() - n0)
    for k in ("img_f", "txt_f", "base_loss", "alignment_loss"):
        assert torch.equal(res[True][0][k], res[False][0][k]), k
    for k in synth.PROMPT_NAMES:
        assert torch.equal(res[True][1][k], res[False][1][k]), k
    saved = res[False][2] - res[True][2]
    print(f"{mode}: {res[False][2]} launches sequential, {res[True][2]} in lock step ({saved} GEMM launches merged)")
    assert saved >= 8 * (CFG.vision_layers - 2), saved
    del enc
    torch.cuda.empty_cache()


def test_packed_text_batch_at_the_benchmarked_configuration(data):
    """bench.py's default text layout — every caption cut at its OWN EOT, the batch packed (engine.PackedIds; 41 instead of 59 rows
    per caption on the synthetic batch) — against the batch cut at the longest caption, bf16, B = 256, depth 3: the same arithmetic per
    live row, so features, losses and factor gradients agree to bf16 kernel-selection noise at most (they are usually the same bits)."""
    from lpi_amd.engine import PackedIds, trim_token_ids
    img, _ = data
    ids_h = synth.token_ids(B)
    ids_t = torch.from_numpy(np.ascontiguousarray(trim_token_ids(ids_h))).to(DEV)
    pk = PackedIds(ids_h).to(DEV)
    assert pk.rows < 0.8 * ids_t.numel()
    enc = DualEncoder(CFG, synth.clip_state_dict(CFG), dtype="bf16", device=DEV)
    res = {}
    for tag, ids in (("trim", ids_t), ("pack", pk)):
        fac = factors()
        out = train_step(enc, img, ids, fac, 3)
        torch.cuda.synchronize()
        res[tag] = ({k: v.clone() for k, v in out.items()}, {k: fac[k].grad.clone() for k in synth.PROMPT_NAMES})
    same = all(torch.equal(res["trim"][0][k], res["pack"][0][k]) for k in ("img_f", "txt_f", "base_loss")) and \
        all(torch.equal(res["trim"][1][k], res["pack"][1][k]) for k in synth.PROMPT_NAMES)
    print(f"packed text batch: {pk.rows / B:.1f} rows per caption (trimmed: {ids_t.shape[1]}); bitwise equal to the trimmed batch: {same}")
    for k in ("img_f", "txt_f"):
        assert float((res["trim"][0][k] - res["pack"][0][k]).abs().max()) < 1e-3, k
    assert abs(float(res["trim"][0]["base_loss"]) - float(res["pack"][0]["base_loss"])) < 1e-3
    for k in synth.PROMPT_NAMES:
        a, b = res["pack"][1][k].double(), res["trim"][1][k].double()
        assert float((a - b).abs().max()) <= 2e-2 * float(b.abs().max()), k
    del enc
    torch.cuda.empty_cache()


@pytest.mark.parametrize("mode,ftol", [("bf16", 5e-3), ("f16", 1.5e-3)])
def test_shared_prefix_text_layout_at_the_benchmarked_configuration(enc32, data, mode, ftol):
    """bench.py's text layout since round 5 — packed AND the 17 positions every caption has in common (SOT + the broadcast context slots) stored once
    (engine.PackedIds(shared=17)) — at B = 256, depth 3: text features bit-identical to the plain packed layout; factor gradients inside the bars the
    throughput modes are held to against the f32 step (which the fixtures pin to the reference), and not further from it than the plain layout's."""
    from lpi_amd.engine import PackedIds
    img, ids = data
    ids_h = synth.token_ids(B)
    f32 = factors()
    o32 = train_step(enc32, img, ids, f32, 3)
    o32 = {k: v.clone() for k, v in o32.items()}
    g32 = {k: f32[k].grad.double().cpu() for k in synth.PROMPT_NAMES}
    enc = DualEncoder(CFG, synth.clip_state_dict(CFG), dtype=mode, device=DEV)
    res = {}
    for tag, shared in (("plain", 0), ("shared", 17)):
        fac = factors()
        pk = PackedIds(ids_h, shared).to(DEV)
        out = train_step(enc, img, pk, fac, 3)
        torch.cuda.synchronize()
        res[tag] = ({k: v.clone() for k, v in out.items()}, {k: fac[k].grad.double().cpu() for k in synth.PROMPT_NAMES}, pk.rows)
    assert res["shared"][2] < 0.65 * res["plain"][2]
    assert torch.equal(res["plain"][0]["img_f"], res["shared"][0]["img_f"]) and torch.equal(res["plain"][0]["txt_f"], res["shared"][0]["txt_f"])
    assert float((res["shared"][0]["txt_f"] - o32["txt_f"]).abs().max()) < ftol
    cos = lambda a, b: float((a * b).sum() / (a.norm() * b.norm()))  # noqa: E731
    rel = lambda a, b: float((a - b).abs().max() / b.abs().max())  # noqa: E731
    report = {k: (cos(res["plain"][1][k], g32[k]), cos(res["shared"][1][k], g32[k]), rel(res["plain"][1][k], g32[k]), rel(res["shared"][1][k], g32[k]))
              for k in synth.PROMPT_NAMES}
    print(f"{mode} factor gradients vs the f32 step (cosine plain, shared; max rel err plain, shared):", {k: tuple(round(x, 5) for x in v) for k, v in report.items()})
    for k, (cp, cs, rp, rs_) in report.items():
        assert cs >= 0.9995 and rs_ <= 0.03, (k, cs, rs_)
        # 6e-3 = the spread of this figure itself: the PLAIN layout's moved 0.0112 -> 0.0035 (f16, dim_1_share) between two builds that differ only in the
        # few-row GEMMs' summation order, the shared layout's 0.0099 -> 0.0083
        assert rs_ <= 1.25 * rp + 6e-3, (k, rp, rs_)
        assert cos(res["shared"][1][k], res["plain"][1][k]) > 0.99995, k
    del enc
    torch.cuda.empty_cache()


@pytest.mark.parametrize("name", ["vitb16_eval", "vitb16_eval12"])
def test_eval_shard_at_vitb16_size_matches_reference(golden, name):
    """north_star: 'R@1 indices bit-identical to reference on a fixed synthetic shard'.  The reference's whole evaluation
    (sprompt.py:433-646: task ids by L1 distance to keys, per-sample prompted features, N_img x N_txt score matrix, per-row rank of the
    best ground truth, R@K) at ViT-B/16 size, through the plugin surface in f32 mode: 32 images x 64 captions x 3 tasks, and (round 6) 256 images x 1 280
    captions x 12 tasks — the whole task pool of a finished continual session.  At 1 280 columns the scores of a random-weight backbone lie 1e-5 apart, so
    besides 'exact wherever the margin allows' every row is held to the sharp bound: a rank can differ from the reference's by at most the number of
    competitors whose reference score is within twice the measured score error of the ground truth's."""
    import json
    import os
    from lpi_amd.retrieval.methods.sprompt import SPrompts
    g = golden(name)
    ret = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lpi_amd", "retrieval")
    args = json.load(open(os.path.join(ret, "configs", "lpi", "coco_lpi.json")))
    args.update(device=[torch.device(DEV)], compute_dtype="f32", num_workers=0, trim_text=True)
    m = SPrompts(args)
    net = m._network.to(torch.device(DEV))
    for t in range(len(net.prompts)):
        for k, v in synth.prompt_factors(9, 16, CFG.vision_width, CFG.transformer_width, task=t).items():
            getattr(net.prompts[t], k).data = torch.from_numpy(v.copy()).to(DEV)
    n_tasks, cpi = int(g["n_tasks"]), int(g["caps_per_img"])
    net.numtask = n_tasks
    m.cur_id = n_tasks - 1
    m.all_keys = [torch.from_numpy(k).to(DEV) for k in g["vkeys"]]
    m.textual_all_keys = [torch.from_numpy(k).to(DEV) for k in g["tkeys"]]
    n_img, n_txt = g["score_i2t"].shape
    img = torch.from_numpy(synth.images(n_img, 224, seed=synth.IMAGE_SEED + 11))

    class DS:
        text = torch.from_numpy(g["token_ids"].astype(np.int64))          # captions as the reference's tokenizer encoded them
        text_cat = list(g["cat_t"])
        img2txt = {i: [cpi * i + j for j in range(cpi)] for i in range(n_img)}
        txt2img = {t: t // cpi for t in range(n_txt)}

    class Loader:
        dataset = DS()

        def __iter__(self):
            for i in range(0, n_img, 16):
                yield img[i:i + 16], torch.arange(i, min(n_img, i + 16)), torch.from_numpy(g["cat_i"][i:i + 16])

    net.eval()
    with torch.no_grad():
        ev = net.extract_vector(img.to(DEV))
        assert float((ev.cpu() - torch.from_numpy(g["extract_vector"])).abs().max()) < 1e-4
        sel_v = m.get_visual_task_id(img.to(DEV)).cpu().numpy()
        sel_t = m.get_textual_task_id(DS.text).cpu().numpy()
    for sel, ref, dist in ((sel_v, g["visual_task_id"], g["visual_task_dist"]), (sel_t, g["textual_task_id"], g["textual_task_dist"])):
        srt = np.sort(dist, 1)
        safe = (srt[:, 1] - srt[:, 0]) > 1e-2
        assert safe.mean() > 0.75 and (sel[safe] == ref[safe]).all()          # integer task ids: exact wherever the choice is not a near-tie
    s_i2t, s_t2i, final_res = m._evaluate_retrieval(Loader())
    err = float(np.abs(s_i2t - g["score_i2t"]).max())
    assert err < 1e-4, err                                                    # cosine scores within the north-star tolerance
    assert np.array_equal(s_t2i, s_i2t.T)
    # rank of the best ground truth per row, bit-exact wherever the reference's margin exceeds 10x the measured score error
    s = torch.cuda.current_stream().cuda_stream
    from lpi_amd import _lib
    for S, gts, ref_r, ref_m, tag in ((s_i2t, [DS.img2txt[i] for i in range(n_img)], g["rank_i2t"], g["rank_margin_i2t"], "i2t"),
                                      (s_t2i, [[DS.txt2img[t]] for t in range(n_txt)], g["rank_t2i"], g["rank_margin_t2i"], "t2i")):
        gt = torch.tensor(gts, dtype=torch.int32, device=DEV)
        r = torch.zeros(len(gts), dtype=torch.int32, device=DEV)
        Sd = torch.from_numpy(np.ascontiguousarray(S)).to(DEV)
        _lib.call("lpi_retrieval_rank", Sd.shape[0], Sd.shape[1], Sd, Sd.shape[1], gt, gt.shape[1], r, s)
        safe = ref_m > 10 * err
        got = r.cpu().numpy()
        # the sharp bound, every row: only competitors whose REFERENCE score is within 2 err of a ground truth's can change sides
        Sref = g["score_i2t"] if tag == "i2t" else g["score_i2t"].T
        close = np.array([max(int((np.abs(Sref[i] - Sref[i, j]) <= 2 * err).sum()) - 1 for j in gts[i]) for i in range(len(gts))])
        print(f"eval shard {name} {tag}: max |score err| {err:.2e}; ranks exact on {int((got == ref_r).sum())} of {len(safe)} rows; asserted exact on "
              f"{int(safe.sum())} (margin > 10 err) and within the competitor count on all (largest count {int(close.max())})")
        if name == "vitb16_eval":
            assert safe.mean() > 0.8
        assert np.array_equal(got[safe], ref_r[safe])
        assert (np.abs(got.astype(np.int64) - ref_r) <= close).all()
        # R@1/5/10 per task (sprompt.py:638-646): a row can change a recall figure only if its rank interval straddles the threshold
        cat = g["cat_i"] if tag == "i2t" else g["cat_t"]
        for t in range(n_tasks):
            rows = cat == t
            for ki, kk in enumerate((1, 5, 10)):
                undecided = int(((ref_r[rows] - close[rows] < kk) & (ref_r[rows] + close[rows] >= kk)).sum())
                refv = g["itm_" + tag][t][ki]
                assert abs(final_res["mscoco"][tag][t][ki] - refv) <= 100.0 * undecided / max(1, int(rows.sum())) + 1e-9, (tag, t, kk)

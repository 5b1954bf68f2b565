"""The text tower's SHARED-PREFIX layout (include/lpi_hip.h: lpi_attn_fwd_shared ..., engine.PackedIds(shared=n)) on a real MI355X.

In the training forward every caption is [SOT][n_ctx context slots][caption][EOT] and the context / deep prompts are broadcast over the batch
(slinet.py:119-130), so under the causal mask (model.py:347-353) the first 1 + n_ctx positions hold the SAME rows for every sample in every block.
The layout stores them once.  Checked here:
  * kernel level — attention forward / backward, the pooled-row attention and the embedding on the shared layout against the SAME kernels on the plain
    packed layout (forward bit for bit; the gradient of the shared rows = the sum of the plain layout's per-sample gradients);
  * engine level — the ViT-B/16 fixture of the reference in both layouts: same features, factor gradients inside the mode's bar, not further from the
    reference than the plain layout; edge cases (one sample, a caption that is only its EOT, 77 positions); the loud errors;
  * plugin level — SliNet's training forward takes the layout, its inference paths (per-sample prompts, un-prompted) do not."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from lpi_amd import _lib, synth  # noqa: E402
from lpi_amd._lib import BF16, F16, F32, call  # noqa: E402
from lpi_amd.engine import DualEncoder, PackedIds  # noqa: E402
from lpi_amd.step import train_step  # noqa: E402

DEV = torch.device("cuda:0")
TDX = {BF16: torch.bfloat16, F16: torch.float16, F32: torch.float32}
PRE = 17


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def stream():
    return torch.cuda.current_stream().cuda_stream


def relerr(got, ref):
    got, ref = got.double().cpu(), ref.double().cpu()
    return float((got - ref).abs().max() / (ref.abs().max() + 1e-30))


def layouts(own, d, seed, dt, width=3):
    """A batch whose samples share their first PRE rows: -> (plain matrix, plain row starts, shared matrix, shared row starts, row maps)."""
    B = len(own)
    pre_rows = rnd(PRE, width * d, seed=seed).to(TDX[dt])
    tails = [rnd(n, width * d, seed=seed + 1 + b).to(TDX[dt]) for b, n in enumerate(own)]
    plain = torch.cat([torch.cat([pre_rows, t]) for t in tails])
    shared = torch.cat([pre_rows] + tails)
    rs_p = np.concatenate([[0], np.cumsum([PRE + n for n in own])])
    rs_s = np.concatenate([[PRE], PRE + np.cumsum(own)])
    return plain, torch.from_numpy(rs_p.astype(np.int32)), shared, torch.from_numpy(rs_s.astype(np.int32)), B


def pad_rows(t, fill=0.0):
    Mp = (t.shape[0] + 255) // 256 * 256
    out = torch.full((Mp, t.shape[1]), fill, dtype=t.dtype)
    out[:t.shape[0]] = t
    return out


# ------------------------------------------------------------------------------------------------ kernels
@pytest.mark.parametrize("dt", [BF16, F16, F32])
@pytest.mark.parametrize("own,H", [([1, 7, 33, 60, 20], 2), ([60], 1), ([3, 3, 16, 15, 48, 1, 31], 8)])
def test_attention_on_the_shared_layout_equals_the_plain_layout(dt, own, H):
    d = H * 64
    plain, rs_p, shared, rs_s, B = layouts(own, d, 100 + H, dt)
    L = PRE + max(own)
    gdt = torch.float32 if dt == F32 else torch.bfloat16      # gradients are bf16 in both 2-byte modes
    gtol = 2e-5 if dt == F32 else 2e-2
    qp, qs = pad_rows(plain).to(DEV), pad_rows(shared).to(DEV)
    Mp_, Ms_ = plain.shape[0], shared.shape[0]
    ctx_p = torch.full((qp.shape[0], d), 3.0, device=DEV, dtype=TDX[dt])
    ctx_s = torch.full((qs.shape[0], d), 3.0, device=DEV, dtype=TDX[dt])
    lse_p = torch.zeros(B, H, L, device=DEV)
    lse_s = torch.zeros(B + 1, H, L, device=DEV)
    call("lpi_attn_fwd_varlen", dt, B, L, rs_p.to(DEV), H, qp, 3 * d, ctx_p, d, lse_p, 1, stream())
    call("lpi_attn_fwd_shared", dt, B, L, rs_s.to(DEV), PRE, H, qs, 3 * d, ctx_s, d, lse_s, stream())
    torch.cuda.synchronize()
    assert bool((ctx_s[Ms_:] == 3.0).all())
    # forward: the same tiles in the same order -> the same bits
    assert torch.equal(ctx_s[:PRE], ctx_p[:PRE]) and torch.equal(lse_s[B, :, :PRE], lse_p[0, :, :PRE])
    for b, n in enumerate(own):
        assert torch.equal(ctx_s[int(rs_s[b]):int(rs_s[b]) + n], ctx_p[int(rs_p[b]) + PRE:int(rs_p[b]) + PRE + n]), b
        assert torch.equal(lse_s[b, :, :n], lse_p[b, :, PRE:PRE + n]), b
        assert torch.equal(ctx_p[int(rs_p[b]):int(rs_p[b]) + PRE], ctx_p[:PRE])          # the premise: the prefix rows ARE the same for every sample

    # backward.  Plain layout: every sample's prefix rows receive their own dctx; the shared rows receive the SUM (what reaches a broadcast prompt).
    dctx_own = [rnd(n, d, seed=300 + b).to(gdt) for b, n in enumerate(own)]
    dctx_pre = [rnd(PRE, d, seed=400 + b, scale=0.5).to(gdt) for b in range(B)]
    dp = pad_rows(torch.cat([torch.cat([dctx_pre[b], dctx_own[b]]) for b in range(B)])).to(DEV)
    pre_sum = torch.stack([t.float() for t in dctx_pre]).sum(0)
    ds = pad_rows(torch.cat([pre_sum.to(gdt)] + dctx_own)).to(DEV)
    dq_p = torch.full((qp.shape[0], 3 * d), 5.0, device=DEV, dtype=gdt)
    dq_s = torch.full((qs.shape[0], 3 * d), 5.0, device=DEV, dtype=gdt)
    del_p, del_s = torch.zeros(B, H, L, device=DEV), torch.zeros(B + 1, H, L, device=DEV)
    scratch = torch.full((B * PRE * 2 * d,), float("nan"), device=DEV)
    call("lpi_attn_bwd_varlen", dt, B, L, rs_p.to(DEV), H, qp, 3 * d, ctx_p, d, dp, d, lse_p, del_p, dq_p, 3 * d, 1, stream())
    call("lpi_attn_bwd_shared", dt, B, L, rs_s.to(DEV), PRE, L, H, qs, 3 * d, ctx_s, d, ds, d, lse_s, del_s, dq_s, 3 * d, scratch, stream())
    torch.cuda.synchronize()
    assert bool((dq_s[Ms_:] == 5.0).all()) and not bool(torch.isnan(scratch).any())
    ref_pre = torch.zeros(PRE, 3 * d, dtype=torch.float64)
    for b, n in enumerate(own):
        r_p, r_s = int(rs_p[b]), int(rs_s[b])
        got, ref = dq_s[r_s:r_s + n], dq_p[r_p + PRE:r_p + PRE + n]
        assert torch.equal(got[:, :d], ref[:, :d]), b                                   # dQ of the own rows: same tiles, same order
        assert relerr(got[:, d:], ref[:, d:]) < gtol, b                                 # dK, dV: the query tiles are cut elsewhere (f32 order, bf16 store)
        assert torch.equal(del_s[b, :, :n], del_p[b, :, PRE:PRE + n])
        ref_pre += dq_p[r_p:r_p + PRE].double().cpu()
    # the shared rows: sum over the samples of the plain layout's prefix-row gradients (dctx of the shared rows was rounded once: bf16 tolerance)
    assert relerr(dq_s[:PRE], ref_pre) < gtol

    # rows_needed = the prefix only (the first block's backward): dQ / dK / dV of the shared rows are complete, delta is there for every row
    dq_s2 = torch.full_like(dq_s, 5.0)
    del_s2 = torch.zeros_like(del_s)
    call("lpi_attn_bwd_shared", dt, B, L, rs_s.to(DEV), PRE, PRE, H, qs, 3 * d, ctx_s, d, ds, d, lse_s, del_s2, dq_s2, 3 * d, scratch, stream())
    torch.cuda.synchronize()
    assert torch.equal(dq_s2[:PRE], dq_s[:PRE]) and torch.equal(del_s2, del_s)


@pytest.mark.parametrize("dt", [BF16, F16, F32])
def test_pooled_attention_on_the_shared_layout(dt):
    own, H = [1, 7, 33, 60, 20, 2], 2
    d = H * 64
    plain, rs_p, shared, rs_s, B = layouts(own, d, 500, dt)
    L = PRE + max(own)
    gdt = torch.float32 if dt == F32 else torch.bfloat16
    qp, qs = pad_rows(plain).to(DEV), pad_rows(shared).to(DEV)
    idx = torch.tensor([PRE + n - 1 for n in own], dtype=torch.int32, device=DEV)      # the EOT's POSITION
    q_rows = rnd(B, d, seed=501).to(TDX[dt]).to(DEV)
    dctx_rows = rnd(B, d, seed=502).to(gdt).to(DEV)
    out = {}
    for tag, qkv, rs, pre in (("p", qp, rs_p, 0), ("s", qs, rs_s, PRE)):
        ctx = torch.zeros(B, d, device=DEV, dtype=TDX[dt])
        lse = torch.zeros(B * H, device=DEV)
        dq = torch.zeros(B, d, device=DEV, dtype=gdt)
        dqkv = torch.full((qkv.shape[0], 3 * d), 7.0, device=DEV, dtype=gdt)
        scratch = torch.full((B * PRE * 2 * d,), float("nan"), device=DEV)
        desc = dict(B=B, L=L, H=H, row_start=rs.to(DEV), q=q_rows, ldq=d, qkv=qkv, ldqkv=3 * d, idx=idx, ctx=ctx, ldctx=d, lse=lse, causal=1, shared_rows=pre)
        _lib.attn_pooled_one(dt, desc, stream())
        desc.update(dctx=dctx_rows, lddctx=d, dq=dq, lddq=d, dqkv=dqkv, lddqkv=3 * d, shared_dkv=scratch if pre else None)
        _lib.attn_pooled_one(dt, desc, stream(), backward=True)
        torch.cuda.synchronize()
        out[tag] = (ctx, lse, dq, dqkv)
        if pre:
            assert not bool(torch.isnan(scratch).any())
    for i in range(3):      # context row, lse, dQ: the same keys in the same order
        assert torch.equal(out["p"][i], out["s"][i]), i
    ref_pre = torch.zeros(PRE, 2 * d, dtype=torch.float64)
    for b, n in enumerate(own):
        r_p, r_s = int(rs_p[b]), int(rs_s[b])
        assert torch.equal(out["s"][3][r_s:r_s + n, d:], out["p"][3][r_p + PRE:r_p + PRE + n, d:]), b
        ref_pre += out["p"][3][r_p:r_p + PRE, d:].double().cpu()
    assert relerr(out["s"][3][:PRE, d:], ref_pre) < (1e-5 if dt == F32 else 1e-2)            # f32 sum of the samples' partials, rounded once
    # the pair form (how the engine issues it beside the vision tower's) = the single form
    ctx2, lse2 = torch.zeros(B, d, device=DEV, dtype=TDX[dt]), torch.zeros(B * H, device=DEV)
    ctx3, lse3 = torch.zeros(B, d, device=DEV, dtype=TDX[dt]), torch.zeros(B * H, device=DEV)
    f = lambda c, l_, rs, qkv, pre: dict(B=B, L=L, H=H, row_start=rs.to(DEV), q=q_rows, ldq=d, qkv=qkv, ldqkv=3 * d, idx=idx, ctx=c, ldctx=d, lse=l_, causal=1,  # noqa: E731
                                         shared_rows=pre)
    _lib.attn_pooled_pair(dt, f(ctx2, lse2, rs_p, qp, 0), f(ctx3, lse3, rs_s, qs, PRE), stream())
    torch.cuda.synchronize()
    assert torch.equal(ctx2, out["p"][0]) and torch.equal(ctx3, out["s"][0]) and torch.equal(lse3, out["s"][1])


def test_text_embedding_on_the_shared_layout():
    B, L, P, d = 6, 40, 16, 128
    g = torch.Generator().manual_seed(9)
    lens = torch.randint(PRE + 1, L + 1, (B,), generator=g)
    lens[0] = L
    ids = torch.randint(1, 1000, (B, L), generator=g)
    ids[:, 0] = 999
    tok, pos, ctxp = rnd(1000, d, seed=1).to(DEV), rnd(77, d, seed=2).to(DEV), rnd(P, d, seed=3).to(DEV)
    rs_p = torch.zeros(B + 1, dtype=torch.int32)
    rs_p[1:] = lens.cumsum(0)
    rs_s = torch.zeros(B + 1, dtype=torch.int32)
    rs_s[0] = PRE
    rs_s[1:] = PRE + (lens - PRE).cumsum(0)
    Mp_, Ms_ = int(rs_p[-1]), int(rs_s[-1])
    xp = torch.full((Mp_, d), 2.0, device=DEV, dtype=torch.float16)
    xs = torch.full((Ms_ + 8, d), 2.0, device=DEV, dtype=torch.float16)
    mp, rp = torch.zeros(Mp_, device=DEV), torch.zeros(Mp_, device=DEV)
    ms, rs_ = torch.zeros(Ms_ + 8, device=DEV), torch.zeros(Ms_ + 8, device=DEV)
    call("lpi_txt_embed_fwd_varlen", F16, B, L, rs_p.to(DEV), P, d, ids.to(DEV), tok, pos, ctxp, 0, xp, mp, rp, stream())
    call("lpi_txt_embed_fwd_shared", F16, B, L, rs_s.to(DEV), PRE, P, d, ids.to(DEV), tok, pos, ctxp, xs, ms, rs_, stream())
    torch.cuda.synchronize()
    assert bool((xs[Ms_:] == 2.0).all())
    assert torch.equal(xs[:PRE], xp[:PRE]) and torch.equal(ms[:PRE], mp[:PRE]) and torch.equal(rs_[:PRE], rp[:PRE])
    for b in range(B):
        n, a, c = int(lens[b]) - PRE, int(rs_s[b]), int(rs_p[b]) + PRE
        assert torch.equal(xs[a:a + n], xp[c:c + n]) and torch.equal(ms[a:a + n], mp[c:c + n]) and torch.equal(rs_[a:a + n], rp[c:c + n]), b
    lib = _lib.load()
    bad = lambda **kw: lib.lpi_txt_embed_fwd_shared(F16, B, L, kw.get("rs", rs_s.to(DEV).data_ptr()), kw.get("pre", PRE), P, d, ids.to(DEV).data_ptr(), tok.data_ptr(),  # noqa: E731
                                                    pos.data_ptr(), kw.get("ctx", ctxp.data_ptr()), xs.data_ptr(), None, None, None)
    assert bad(rs=None) == -22 and bad(pre=16) == -22 and bad(ctx=None) == -22


@pytest.mark.parametrize("dt", [BF16, F32])
@pytest.mark.parametrize("B,H", [(1, 1), (5, 2), (16, 8), (33, 12), (256, 8)])
def test_shared_kv_reduce_against_a_float64_sum(dt, B, H):
    """dqkv[key][K | V columns] (+)= sum over the samples of the f32 partials: against the f64 sum (f32 output: a few ulps of the fixed summation tree; bf16
    output: one rounding on top), the Q columns and the rows behind the shared ones untouched, twice the same bits."""
    d = H * 64
    part = rnd(B, PRE, 2 * d, seed=7 * B + H).to(DEV)
    base = rnd(64, 3 * d, seed=11).to(TDX[dt])
    ref = part.double().sum(0).cpu()
    for acc in (0, 1):
        outs = []
        for _ in range(2):
            dq = base.clone().to(DEV)
            call("lpi_shared_kv_reduce", dt, B, PRE, H, part, dq, 3 * d, acc, stream())
            torch.cuda.synchronize()
            outs.append(dq.cpu())
        assert torch.equal(outs[0], outs[1])
        want = ref + (base[:PRE, d:].double() if acc else 0.0)
        err = (outs[0][:PRE, d:].double() - want).abs().max() / want.abs().max()
        assert float(err) < (2e-6 if dt == F32 else 6e-3), (acc, float(err))
        assert torch.equal(outs[0][:PRE, :d], base[:PRE, :d]) and torch.equal(outs[0][PRE:], base[PRE:])


def test_shared_entry_points_refuse_what_they_cannot_do():
    lib = _lib.load()
    B, L, H, d = 2, 40, 1, 64
    q = torch.zeros(256, 3 * d, device=DEV, dtype=torch.bfloat16)
    c = torch.zeros(256, d, device=DEV, dtype=torch.bfloat16)
    lse = torch.zeros(B + 1, H, L, device=DEV)
    rs = torch.tensor([PRE, PRE + 10, PRE + 33], dtype=torch.int32, device=DEV)
    fwd = lambda dt=BF16, rs_=rs.data_ptr(), pre=PRE: lib.lpi_attn_fwd_shared(dt, B, L, rs_, pre, H, q.data_ptr(), 3 * d, c.data_ptr(), d, lse.data_ptr(), None)  # noqa: E731
    assert fwd() == 0
    assert fwd(rs_=None) == -22 and fwd(pre=0) == -22 and fwd(pre=L) == -22
    scratch = torch.zeros(B * PRE * 2 * d, device=DEV)
    bwd = lambda need=L, sc=scratch.data_ptr(): lib.lpi_attn_bwd_shared(BF16, B, L, rs.data_ptr(), PRE, need, H, q.data_ptr(), 3 * d, c.data_ptr(), d, c.data_ptr(), d,  # noqa: E731
                                                                        lse.data_ptr(), lse.data_ptr(), q.data_ptr(), 3 * d, sc, None)
    assert bwd(need=PRE - 1) == -22 and bwd(sc=None) == -22
    torch.cuda.synchronize()


# ------------------------------------------------------------------------------------------------ engine
def _factors(cfg):
    f = synth.prompt_factors(9, 16, cfg.vision_width, cfg.transformer_width)
    return {k: torch.from_numpy(v).to(DEV).requires_grad_(True) for k, v in f.items()}


def _step(enc, cfg, batch, ids, depth, shared):
    fac = _factors(cfg)
    img = torch.from_numpy(synth.images(batch, cfg.image_resolution)).to(DEV)
    out = train_step(enc, img, PackedIds(ids, shared).to(DEV), fac, depth)
    torch.cuda.synchronize()
    res = {k: v.float().cpu().numpy() for k, v in out.items()}
    for k in synth.PROMPT_NAMES:
        res["grad." + k] = fac[k].grad.cpu().numpy()
    return res


def _maxerr(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())


def _cos(a, b):
    return float((a * b).sum() / np.sqrt((a * a).sum() * (b * b).sum()))


@pytest.mark.parametrize("name,cfgname,batch,depth", [("tiny_d1", "tiny", 4, 1), ("tiny_d2_patched", "tiny", 4, 2), ("vitb16_d3_patched", "ViT-B/16", 8, 3)])
def test_f32_step_on_the_shared_layout_meets_the_parity_bar_of_the_reference_fixtures(golden, name, cfgname, batch, depth):
    """The layout is EXACT, and in the f32 parity mode that shows at the reference's own bar: the step on PackedIds(shared=17) against the fixtures captured from
    the imported reference — features / logits / losses within 1e-4, prompt-factor gradients within 1e-3 relative (tests/test_model_gpu.py holds the plain
    layout to the same numbers)."""
    cfg = synth.CONFIGS[cfgname]
    g = golden(name)
    enc = DualEncoder(cfg, synth.clip_state_dict(cfg), dtype="f32", device=DEV)
    res = _step(enc, cfg, batch, g["token_ids"], depth, PRE)
    assert enc.txt._ws[(batch, True)]["pre"] == PRE
    logits = enc.logit_scale_exp * res["img_f"] @ res["txt_f"].T
    assert _maxerr(res["img_f"], g["img_f"]) <= 1e-4 and _maxerr(res["txt_f"], g["txt_f"]) <= 1e-4 and _maxerr(logits, g["logits"]) <= 1e-4
    for k in ("base_loss", "alignment_loss"):
        assert abs(float(res[k]) - float(g[k])) <= 1e-4, k
    worst = 0.0
    for k in synth.PROMPT_NAMES:
        e, scale = _maxerr(res["grad." + k], g["grad." + k]), float(np.abs(g["grad." + k]).max())
        worst = max(worst, e / scale)
        assert e <= 1e-3 * scale + 1e-7, (k, e, scale)
    print(f"{name}: f32 on the shared layout vs the reference fixture: logits {_maxerr(logits, g['logits']):.2e}, factor gradients {worst:.2e} relative")


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_vitb16_fixture_in_both_layouts(golden, dtype):
    """The reference's own outputs (tests/golden/vitb16_d3_patched.npz: ViT-B/16, 8 pairs, prompt depth 3) against the step in the plain packed layout and
    in the shared-prefix layout: the shared layout is not further from the reference, and the two agree far inside the mode's error."""
    cfg = synth.VIT_B16
    g = golden("vitb16_d3_patched")
    enc = DualEncoder(cfg, synth.clip_state_dict(cfg), dtype=dtype, device=DEV)
    n0 = _lib.launch_count()
    plain = _step(enc, cfg, 8, g["token_ids"], 3, 0)
    n1 = _lib.launch_count()
    shared = _step(enc, cfg, 8, g["token_ids"], 3, PRE)
    n2 = _lib.launch_count()
    assert enc.txt._ws[(8, True)]["pre"] == PRE
    assert (n2 - n1) - (n1 - n0) == cfg.transformer_layers          # one lpi_shared_kv_reduce per block, nothing else
    grads = ["grad." + k for k in synth.PROMPT_NAMES]
    assert np.array_equal(plain["img_f"], shared["img_f"])
    ep, es = _maxerr(plain["txt_f"], g["txt_f"]), _maxerr(shared["txt_f"], g["txt_f"])
    print(f"{dtype}: text feature error vs the reference: plain {ep:.2e}, shared {es:.2e}; between them {_maxerr(plain['txt_f'], shared['txt_f']):.2e}")
    assert es <= 1.25 * ep + 1e-5
    assert _maxerr(plain["txt_f"], shared["txt_f"]) <= (3e-3 if dtype == "bf16" else 6e-4)
    for k in ("base_loss", "alignment_loss"):
        assert abs(float(shared[k]) - float(g[k])) <= abs(float(plain[k]) - float(g[k])) + 2e-3
    rel = lambda a, b: _maxerr(a, b) / (float(np.abs(b).max()) + 1e-30)  # noqa: E731
    rows = {k: (round(_cos(plain[k], g[k]), 5), round(_cos(shared[k], g[k]), 5), round(rel(plain[k], g[k]), 4), round(rel(shared[k], g[k]), 4)) for k in grads}
    print(f"{dtype}: factor gradients vs the reference (cosine plain, cosine shared, max rel err plain, shared): {rows}")
    for k in grads:
        assert _cos(shared[k], g[k]) > (0.998 if dtype == "bf16" else 0.999), k            # the bars of tests/test_model_gpu.py for the plain layout
        assert rel(shared[k], g[k]) <= (8e-2 if dtype == "bf16" else 5e-2), k
        assert rel(shared[k], g[k]) <= 1.3 * rel(plain[k], g[k]) + 5e-3, k                  # ... and not further from the reference than it
        assert _cos(shared[k], plain[k]) > 0.9995, k


def test_towers_one_after_the_other_equal_the_lock_stepped_towers_on_the_shared_layout():
    """run_alone issues lpi_attn_fwd_shared / lpi_attn_pooled_*_desc / single launches where run_lockstep issues the pair forms: the same kernel bodies,
    the same bits (ViT-B/16 widths at 8 pairs, so that the paired launches really pair)."""
    cfg = synth.VIT_B16
    enc = DualEncoder(cfg, synth.clip_state_dict(cfg), dtype="bf16", device=DEV)
    ids = synth.token_ids(8)
    img = torch.from_numpy(synth.images(8, cfg.image_resolution)).to(DEV)
    res = {}
    for lock in (True, False):
        fac = _factors(cfg)
        out = train_step(enc, img, PackedIds(ids, PRE).to(DEV), fac, 3, lockstep=lock)
        torch.cuda.synchronize()
        res[lock] = (out["txt_f"].clone(), out["img_f"].clone(), {k: fac[k].grad.clone() for k in synth.PROMPT_NAMES})
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    for k in synth.PROMPT_NAMES:
        assert torch.equal(res[True][2][k], res[False][2][k]), k


def test_text_micro_batches_on_stream_lanes_keep_the_shared_layout():
    """overlap_towers with two text lanes: PackedIds[lo:hi] keeps shared = 17 (each micro-batch stores its own copy of the common rows) and the step agrees
    with the lock-stepped one (other row counts pick other GEMM kernels: bf16 round-off, not bits)."""
    cfg = synth.TINY
    enc = DualEncoder(cfg, synth.clip_state_dict(cfg), dtype="bf16", device=DEV)
    ids = synth.token_ids(8)
    img = torch.from_numpy(synth.images(8, cfg.image_resolution)).to(DEV)
    pk = PackedIds(ids, PRE)
    assert pk[0:4].shared == PRE and pk[4:8].rows == PRE + int((pk.lengths[4:8] - PRE).sum())
    res = {}
    for tag, kw in (("lock", {}), ("lanes", dict(overlap_towers=True, text_lanes=2))):
        fac = _factors(cfg)
        out = train_step(enc, img, PackedIds(ids, PRE).to(DEV) if tag == "lock" else PackedIds(ids, PRE), fac, 2, **kw)
        torch.cuda.synchronize()
        res[tag] = (out["txt_f"].float().cpu(), {k: fac[k].grad.cpu() for k in synth.PROMPT_NAMES})
    assert float((res["lock"][0] - res["lanes"][0]).abs().max()) < 4e-3
    for k in synth.PROMPT_NAMES:
        a, b = res["lock"][1][k], res["lanes"][1][k]
        assert float((a - b).abs().max()) <= 3e-2 * float(a.abs().max()) + 1e-6, k


def _ids_with_lengths(lengths, seed=3):
    """[B, 77] token ids: SOT, 16 placeholder slots, caption tokens, EOT at position lengths[b] - 1."""
    g = np.random.default_rng(seed)
    ids = np.zeros((len(lengths), 77), dtype=np.int64)
    for b, n in enumerate(lengths):
        ids[b, 0] = 49406
        ids[b, 1:17] = 343
        ids[b, 17:n - 1] = g.integers(1000, 40000, n - 1 - 17)
        ids[b, n - 1] = 49407
    return ids


@pytest.mark.parametrize("lengths", [[77], [18, 77, 18, 30], [18, 18, 18], [19, 48, 49, 50, 77, 33, 20, 64, 65]])
def test_engine_edge_cases_of_the_shared_layout(lengths):
    """One sample; captions that are only their EOT behind the context slots; 77 positions; own lengths around the 32-row tile edges: both layouts
    give the same text features (bf16 mode: the same forward arithmetic) and close gradients."""
    cfg = synth.TINY
    enc = DualEncoder(cfg, synth.clip_state_dict(cfg), dtype="bf16", device=DEV)
    ids = _ids_with_lengths(lengths)
    B = len(lengths)
    plain = _step(enc, cfg, B, ids, 2, 0)
    shared = _step(enc, cfg, B, ids, 2, PRE)
    assert np.isfinite(shared["txt_f"]).all()
    assert _maxerr(plain["txt_f"], shared["txt_f"]) < 4e-3           # (different GEMM kernels at these row counts: bf16 round-off, not bits)
    for k in synth.PROMPT_NAMES:
        a, b = plain["grad." + k], shared["grad." + k]
        assert np.isfinite(b).all()
        assert _maxerr(a, b) <= 3e-2 * np.abs(a).max() + 1e-6, k


def test_the_layout_is_refused_where_the_rows_are_not_shared():
    cfg = synth.TINY
    enc = DualEncoder(cfg, synth.clip_state_dict(cfg), dtype="bf16", device=DEV)
    ids = synth.token_ids(4)
    sh = PackedIds(ids, PRE).to(DEV)
    pr = torch.zeros(9, 16, cfg.transformer_width, device=DEV)
    with pytest.raises(ValueError, match="broadcast prompts"):
        enc.encode_text(sh, None)                                                  # un-prompted (extract_textual_vector)
    with pytest.raises(ValueError, match="broadcast prompts"):
        enc.encode_text(sh, pr[None].repeat(4, 1, 1, 1))                           # per-sample stacks (textual_interface)
    with pytest.raises(ValueError, match="broadcast prompts"):
        enc.encode_text(sh, pr, use_ctx=False)
    with pytest.raises(ValueError, match="broadcast prompts"):
        enc.encode_text(PackedIds(ids, 9).to(DEV), pr)
    a = enc.encode_text(sh, pr, 2)                                                 # fine: broadcast prompts spliced in
    b = enc.encode_text(PackedIds(ids).to(DEV), pr, 2)
    assert float((a - b).abs().max()) < 4e-3
    f32 = DualEncoder(cfg, synth.clip_state_dict(cfg), dtype="f32", device=DEV)       # the parity mode takes the layout too (its own test below)
    c = f32.encode_text(sh, pr.float(), 2)
    e = f32.encode_text(PackedIds(ids).to(DEV), pr.float(), 2)
    assert float((c - e).abs().max()) < 1e-5
    short = _ids_with_lengths([18, 30])
    short[0, 16], short[0, 17] = 49407, 0                    # EOT inside the context slots: nothing behind the shared positions
    with pytest.raises(ValueError, match="continue behind"):
        PackedIds(short, PRE)
    bad = synth.token_ids(3).copy()
    bad[1, 0] -= 1
    with pytest.raises(ValueError, match="same token"):
        PackedIds(bad, PRE)


# ------------------------------------------------------------------------------------------------ plugin
def test_slinet_trains_on_the_shared_layout_and_evaluates_on_the_plain_one(tmp_path, monkeypatch):
    from lpi_amd import synth_bpe
    from lpi_amd.retrieval.models.clip import prompt_learner as PL
    from lpi_amd.retrieval.models.slinet import SliNet
    monkeypatch.setenv("LPI_BPE_VOCAB", synth_bpe.write_table(tmp_path / "bpe.txt.gz", seed=1))
    monkeypatch.setattr(PL, "_tokenizer", None)
    ret = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lpi_amd", "retrieval")
    args = json.load(open(os.path.join(ret, "configs", "lpi", "coco_lpi.json")))
    args.update(backbonename="tiny", visual_dim=128, textual_dim=128, device=[DEV], compute_dtype="bf16", batch_size=4, epochs=1, num_workers=0)
    net = SliNet(args).to(DEV)
    for t in range(len(net.prompts)):
        for k, v in synth.prompt_factors(9, 16, 128, 128, task=t).items():
            getattr(net.prompts[t], k).data = torch.from_numpy(v.copy()).to(DEV)
    net.numtask = 1
    net.train()
    img = torch.from_numpy(synth.images(4, 32)).to(DEV)
    caps = ["a dog on a sofa", "two people riding bikes down a street", "cat", "a plate of food with a fork and a knife on a wooden table"]
    assert net._shared_rows() == PRE and net.prepare_text(caps).shared == PRE
    out = net.train_step(img, caps)
    torch.cuda.synchronize()
    assert net.engine.txt._ws[(4, True)]["pre"] == PRE
    g_shared = {k: getattr(net.prompts[0], k).grad.clone() for k in synth.PROMPT_NAMES}
    net.args["share_text_prefix"] = False
    assert net.prepare_text(caps).shared == 0
    out2 = net.train_step(img, caps)
    torch.cuda.synchronize()
    assert net.engine.txt._ws[(4, True)]["pre"] == 0
    assert float((out["text_features"] - out2["text_features"]).abs().max()) < 4e-3
    for k in synth.PROMPT_NAMES:
        a, b = getattr(net.prompts[0], k).grad, g_shared[k]
        assert float((a - b).abs().max()) <= 3e-2 * float(a.abs().max()) + 1e-6, k
    # inference: per-sample stacks / un-prompted features run on the plain layout whatever the training forward used
    net.args["share_text_prefix"] = True
    net.eval()
    with torch.no_grad():
        t0 = net.extract_textual_vector(caps)
        sel = torch.zeros(4, dtype=torch.long, device=DEV)
        t1 = net.textual_interface(caps, sel) if hasattr(net, "textual_interface") else None
    assert torch.isfinite(t0).all() and (t1 is None or torch.isfinite(t1).all())
    f32 = SliNet(dict(args, compute_dtype="f32"))
    assert f32._shared_rows() == PRE and SliNet(dict(args, share_text_prefix=False))._shared_rows() == 0
    monkeypatch.setattr(PL, "_tokenizer", None)

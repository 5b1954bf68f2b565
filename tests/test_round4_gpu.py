"""Round-4 additions on a real MI355X, all through the C ABI:
  * lpi_row_jobs / the pair launches of the step's tail against the single-op entry points they replace — bit for bit;
  * the fused loss / DecomposedPrompt / alignment kernels against the multi-launch forms — bit for bit — and against f64;
  * the launch count of a steady-state training step (the tail is paired: fewer launches, no ATen kernel), asserted;
  * a prompt gradient held across two steps does not change (the persistent buffer is never handed to autograd's users);
  * FlatSGD refuses to step parameters that were re-seated behind its back;
  * lock-stepped towers whose conditional requests differ re-align (one tower at LPI_ROWSTATS = 0);
  * the accuracy envelope of the single-sweep LayerNorm statistics (rows with a large mean, 'massive activation' channels)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from lpi_amd import _lib, engine as E, synth  # noqa: E402
from lpi_amd._lib import BF16, F16, F32, call  # noqa: E402

DEV = "cuda:0"


@pytest.fixture(scope="module")
def golden():
    import os

    def load(name):
        return dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"), allow_pickle=False))
    return load


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def stream():
    return torch.cuda.current_stream().cuda_stream


# ------------------------------------------------------------------------------------------------ row jobs
def test_row_jobs_equal_the_single_op_launches_bit_for_bit():
    B, L, dv, dt_, P = 5, 21, 768, 512, 4
    s = stream()
    n0 = _lib.launch_count()
    jobs, checks = [], []
    for d, seed in ((dv, 1), (dt_, 2)):
        # POOL_LN_FWD (fp16 stream -> bf16 rows, with the raw gather) vs lpi_pool_ln_fwd + lpi_gather_rows
        x = rnd(B * L, d, seed=seed).half().to(DEV)
        idx = torch.randint(0, L, (B,), generator=torch.Generator().manual_seed(seed)).int().to(DEV)
        gam, bet = (1 + 0.1 * rnd(d, seed=seed + 10)).to(DEV), (0.1 * rnd(d, seed=seed + 11)).to(DEV)
        y0, y1 = torch.zeros(B, d, dtype=torch.bfloat16, device=DEV), torch.zeros(B, d, dtype=torch.bfloat16, device=DEV)
        st0, st1 = torch.zeros(2, B, device=DEV), torch.zeros(2, B, device=DEV)
        raw0, raw1 = torch.zeros(B, d, device=DEV), torch.zeros(B, d, device=DEV)
        call("lpi_pool_ln_fwd", BF16, F16, B, L, d, x, idx, gam, bet, y0, d, st0[0], st0[1], s)
        call("lpi_gather_rows", F16, B, L, d, x, idx, raw0, s)
        jobs.append(_lib.row_job(_lib.ROWOP_POOL_LN_FWD, B=B, L=L, d=d, dt_a=F16, dt_b=BF16, a=x, idx=idx, gamma=gam, beta=bet, out=y1, ld_c=d, mean=st1[0],
                                 rstd=st1[1], out2=raw1))
        checks += [(y0, y1), (st0, st1), (raw0, raw1)]
    _lib.row_jobs(jobs, s)
    # L2NORM_FWD / L2NORM_BWD (+ bf16 copy) vs lpi_l2norm_fwd / lpi_l2norm_bwd + lpi_cast
    Ed = 512
    jobs = []
    for seed in (3, 4):
        f = rnd(B, Ed, seed=seed).to(DEV)
        o0, o1, i0, i1 = torch.zeros(B, Ed, device=DEV), torch.zeros(B, Ed, device=DEV), torch.zeros(B, device=DEV), torch.zeros(B, device=DEV)
        call("lpi_l2norm_fwd", B, Ed, f, Ed, o0, Ed, i0, s)
        jobs.append(_lib.row_job(_lib.ROWOP_L2NORM_FWD, B=B, d=Ed, a=f, ld_a=Ed, out=o1, ld_c=Ed, mean=i1))
        checks += [(o0, o1), (i0, i1)]
        g = rnd(B, Ed, seed=seed + 20).to(DEV)
        d0, d1 = torch.zeros(B, Ed, device=DEV), torch.zeros(B, Ed, device=DEV)
        c0, c1 = torch.zeros(B, Ed, dtype=torch.bfloat16, device=DEV), torch.zeros(B, Ed, dtype=torch.bfloat16, device=DEV)
        call("lpi_l2norm_bwd", B, Ed, o0, Ed, g, Ed, i0, d0, Ed, s)
        call("lpi_cast", F32, BF16, d0.numel(), d0, c0, s)
        jobs.append(_lib.row_job(_lib.ROWOP_L2NORM_BWD, B=B, d=Ed, a=o0, ld_a=Ed, b=g, ld_b=Ed, mean_in=i0, out=d1, ld_c=Ed, out2=c1, dt_b=BF16))
        checks += [(d0, d1), (c0, c1)]
    _lib.row_jobs(jobs, s)
    # POOL_LN_BWD, LN_BWD (f32 rows, bf16 dy / copy, accumulate), SCATTER_ADD
    jobs = []
    for d, seed in ((dv, 5), (dt_, 6)):
        dy = rnd(B, d, seed=seed).to(DEV)
        x = rnd(B, d, seed=seed + 1).to(DEV)
        gam = (1 + 0.1 * rnd(d, seed=seed + 2)).to(DEV)
        mean, rstd = x.mean(1).contiguous(), (1.0 / (x.var(1, unbiased=False) + 1e-5).sqrt()).contiguous()
        a0, a1 = torch.zeros(B, d, device=DEV), torch.zeros(B, d, device=DEV)
        b0, b1 = torch.zeros(B, d, dtype=torch.bfloat16, device=DEV), torch.zeros(B, d, dtype=torch.bfloat16, device=DEV)
        call("lpi_pool_ln_bwd", BF16, B, 1, d, dy, d, x, None, gam, mean, rstd, a0, b0, s)
        jobs.append(_lib.row_job(_lib.ROWOP_POOL_LN_BWD, B=B, L=1, d=d, dt_b=BF16, a=dy, ld_a=d, b=x, gamma=gam, mean_in=mean, rstd_in=rstd, out=a1, out2=b1))
        checks += [(a0, a1), (b0, b1)]
        dyb = dy.to(torch.bfloat16)
        e0, e1 = rnd(B, d, seed=seed + 3).to(DEV), rnd(B, d, seed=seed + 3).to(DEV)
        f0, f1 = torch.zeros(B, d, dtype=torch.bfloat16, device=DEV), torch.zeros(B, d, dtype=torch.bfloat16, device=DEV)
        call("lpi_layernorm_bwd", BF16, BF16, F32, B, d, dyb, d, x, d, gam, mean, rstd, e0, d, f0, d, 1, s)
        jobs.append(_lib.row_job(_lib.ROWOP_LN_BWD, B=B, d=d, dt_a=BF16, dt_b=BF16, a=dyb, ld_a=d, b=x, ld_b=d, gamma=gam, mean_in=mean, rstd_in=rstd, out=e1,
                                 out2=f1, ld_c=d, flag=1))
        checks += [(e0, e1), (f0, f1)]
    _lib.row_jobs(jobs, s)
    jobs = []
    for d, seed in ((dv, 7), (dt_, 8)):
        src = rnd(B, d, seed=seed).to(torch.bfloat16).to(DEV)
        idx = torch.randint(0, L, (B,), generator=torch.Generator().manual_seed(seed)).int().to(DEV)
        t0 = rnd(B * L, d, seed=seed + 1).to(torch.bfloat16).to(DEV)
        t1 = t0.clone()
        call("lpi_scatter_add_rows", BF16, B, L, d, src, d, idx, t0, d, s)
        jobs.append(_lib.row_job(_lib.ROWOP_SCATTER_ADD, B=B, L=L, d=d, dt_a=BF16, a=src, ld_a=d, idx=idx, out=t1, ld_c=d))
        checks.append((t0, t1))
        # PROMPT_ADD on a ragged batch with statistics
        lens = torch.tensor([12, 7, 9, 21, 6])
        rs_ = torch.cat([torch.zeros(1, dtype=torch.long), lens.cumsum(0)]).int().to(DEV)
        rows = int(lens.sum())
        xp0 = rnd(rows, d, seed=seed + 2).half().to(DEV)
        xp1 = xp0.clone()
        pr = rnd(P, d, seed=seed + 3).to(DEV)
        sp0, sp1 = torch.zeros(2, rows, device=DEV), torch.zeros(2, rows, device=DEV)
        call("lpi_prompt_add_varlen", F16, B, L, rs_, P, d, xp0, pr, 0, sp0[0], sp0[1], s)
        jobs.append(_lib.row_job(_lib.ROWOP_PROMPT_ADD, B=B, L=L, row_start=rs_, P=P, d=d, dt_a=F16, out=xp1, a=pr, bstride=0, mean=sp1[0], rstd=sp1[1]))
        checks += [(xp0, xp1), (sp0, sp1)]
    _lib.row_jobs(jobs, s)
    # GATHER_BATCH_ROWS, LN_BWD_ROWS_H16, VIS_PROMPT_ROWS_BWD
    jobs = []
    for d, seed in ((dv, 9), (dt_, 10)):
        src = rnd(B * L, 3 * d, seed=seed).to(torch.bfloat16).to(DEV)
        g0, g1 = torch.zeros(B * P, 3 * d, dtype=torch.bfloat16, device=DEV), torch.zeros(B * P, 3 * d, dtype=torch.bfloat16, device=DEV)
        call("lpi_gather_batch_rows_varlen", BF16, B, L, None, 1, P, 3 * d, src, 3 * d, g0, 3 * d, s)
        jobs.append(_lib.row_job(_lib.ROWOP_GATHER_BATCH_ROWS, B=B, L=L, row0=1, P=P, d=3 * d * 2 // 16, a=src, ld_a=3 * d * 2 // 16, out=g1, ld_c=3 * d * 2 // 16))
        checks.append((g0, g1))
        dyc = rnd(B * P, d, seed=seed + 1).to(torch.bfloat16).to(DEV)
        x = rnd(B * L, d, seed=seed + 2).half().to(DEV)
        gam = (1 + 0.1 * rnd(d, seed=seed + 3)).to(DEV)
        xf = x.float()
        mean, rstd = xf.mean(1).contiguous(), (1.0 / (xf.var(1, unbiased=False) + 1e-5).sqrt()).contiguous()
        s0 = rnd(B * L, d, seed=seed + 4).to(torch.bfloat16).to(DEV)
        s1 = s0.clone()
        call("lpi_layernorm_bwd_rows_varlen", BF16, BF16, F16, B, L, None, 1, P, d, dyc, d, x, d, gam, mean, rstd, None, 0, s0, d, 1, s)
        jobs.append(_lib.row_job(_lib.ROWOP_LN_BWD_ROWS_H16, B=B, L=L, row0=1, P=P, d=d, a=dyc, ld_a=d, b=x, ld_b=d, gamma=gam, mean_in=mean, rstd_in=rstd, out2=s1,
                                 ld_c=d, flag=1))
        checks.append((s0, s1))
    _lib.row_jobs(jobs, s)
    G2 = L - 1 - P
    pr = rnd(P, dv, seed=30).to(DEV)
    gam = (1 + 0.1 * rnd(dv, seed=31)).to(DEV)
    mean, rstd = rnd(B * L, seed=32).to(DEV) * 0.1, (1 + 0.1 * rnd(B * L, seed=33)).abs().to(DEV)
    v0 = rnd(B * L, dv, seed=34).to(torch.bfloat16).to(DEV)
    v1 = v0.clone()
    dp0, dp1 = torch.zeros(P, dv, device=DEV), torch.zeros(P, dv, device=DEV)
    call("lpi_vis_assemble_bwd", BF16, B, G2, P, dv, v0, pr, 0, gam, mean, rstd, dp0, s)
    _lib.row_jobs([_lib.row_job(_lib.ROWOP_VIS_PROMPT_ROWS_BWD, B=B, L=L, P=P, d=dv, dt_a=BF16, out=v1, a=pr, bstride=0, gamma=gam, mean_in=mean, rstd_in=rstd)], s)
    # ... and the batch sums of two towers as one launch (one of them accumulating)
    w0 = rnd(B * L, dt_, seed=35).to(torch.bfloat16).to(DEV)
    q0, q1 = torch.ones(P, dt_, device=DEV), torch.ones(P, dt_, device=DEV)
    call("lpi_rows_sum_over_batch_varlen", BF16, B, L, None, 1, P, dt_, w0, q0, 1, s)
    _lib.rows_sum_pair(BF16, (B, L, None, 1, P, dv, v1, dp1, 0), (B, L, None, 1, P, dt_, w0, q1, 1), s)
    checks += [(v0, v1), (dp0, dp1), (q0, q1)]
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(checks):
        assert torch.equal(a, b), i
    assert _lib.launch_count() - n0 < 60
    with pytest.raises(_lib.LpiError):       # an invalid job is refused before anything is launched
        _lib.row_jobs([_lib.row_job(_lib.ROWOP_L2NORM_FWD, B=B, d=Ed, a=None, ld_a=Ed, out=o1, ld_c=Ed, mean=i1)], s)


@pytest.mark.parametrize("dt", [F32, BF16, F16])
def test_pooled_attention_pair_launch_equals_two_launches(dt):
    td = {F32: torch.float32, BF16: torch.bfloat16, F16: torch.float16}[dt]
    gd = torch.float32 if dt == F32 else torch.bfloat16      # gradients are bf16 after an f16 forward
    s = stream()
    probs = []
    for (B, L, H, causal, seed) in ((3, 213, 3, 0, 1), (4, 59, 2, 1, 2)):
        d = 64 * H
        q = rnd(B, d, seed=seed).to(td).to(DEV)
        qkv = rnd(B * L, 3 * d, seed=seed + 1).to(td).to(DEV)
        idx = torch.randint(1, L, (B,), generator=torch.Generator().manual_seed(seed)).int().to(DEV) if causal else None
        probs.append(dict(B=B, L=L, H=H, d=d, q=q, qkv=qkv, idx=idx, causal=causal, dctx=rnd(B, d, seed=seed + 2).to(gd).to(DEV)))
    outs = []
    for mode in ("single", "pair"):
        res = []
        for p in probs:
            B, L, H, d = p["B"], p["L"], p["H"], p["d"]
            res.append(dict(ctx=torch.zeros(B, d, dtype=td, device=DEV), lse=torch.zeros(B * H, device=DEV), dq=torch.zeros(B, d, dtype=gd, device=DEV),
                            dqkv=torch.zeros(B * L, 3 * d, dtype=gd, device=DEV)))
        if mode == "single":
            for p, r in zip(probs, res):
                call("lpi_attn_pooled_fwd_varlen", dt, p["B"], p["L"], None, p["H"], p["q"], p["d"], p["qkv"], 3 * p["d"], p["idx"], r["ctx"], p["d"], r["lse"],
                     p["causal"], s)
                call("lpi_attn_pooled_bwd_varlen", dt, p["B"], p["L"], None, p["H"], p["q"], p["d"], p["qkv"], 3 * p["d"], p["idx"], p["dctx"], p["d"], r["lse"],
                     r["dq"], p["d"], r["dqkv"], 3 * p["d"], p["causal"], s)
        else:
            f = [dict(B=p["B"], L=p["L"], H=p["H"], row_start=None, q=p["q"], ldq=p["d"], qkv=p["qkv"], ldqkv=3 * p["d"], idx=p["idx"], ctx=r["ctx"], ldctx=p["d"],
                      lse=r["lse"], causal=p["causal"]) for p, r in zip(probs, res)]
            _lib.attn_pooled_pair(dt, f[0], f[1], s)
            b = [dict(B=p["B"], L=p["L"], H=p["H"], row_start=None, q=p["q"], ldq=p["d"], qkv=p["qkv"], ldqkv=3 * p["d"], idx=p["idx"], dctx=p["dctx"],
                      lddctx=p["d"], lse=r["lse"], dq=r["dq"], lddq=p["d"], dqkv=r["dqkv"], lddqkv=3 * p["d"], causal=p["causal"]) for p, r in zip(probs, res)]
            _lib.attn_pooled_pair(dt, b[0], b[1], s, backward=True)
        outs.append(res)
    torch.cuda.synchronize()
    for r0, r1 in zip(*outs):
        for k in r0:
            assert torch.equal(r0[k], r1[k]), k


# ------------------------------------------------------------------------------------------------ fused loss / CP / alignment kernels
@pytest.mark.parametrize("n,r0,nloc", [(256, 0, 256), (300, 44, 200), (2048, 512, 256)])
def test_clip_loss_local_two_launches_equal_the_four(n, r0, nloc):
    s = stream()
    lg = (rnd(n, n, seed=n) * 3).to(DEV)
    l0, l1 = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    lse0, lse1 = torch.zeros(2, n, device=DEV), torch.zeros(2, n, device=DEV)
    g0, gt0, g1, gt1 = (torch.zeros(nloc, n, device=DEV) for _ in range(4))
    call("lpi_clip_loss_fwd_bwd", n, lg, n, 1.0, l0, None, n, lse0[0], lse0[1], s)
    call("lpi_clip_loss_local_grad", n, lg, n, lse0[0], lse0[1], 1.0, r0, nloc, g0, gt0, n, s)
    n0 = _lib.launch_count()
    call("lpi_clip_loss_local", n, lg, n, 1.0, r0, nloc, l1, lse1[0], lse1[1], g1, gt1, n, s)
    assert _lib.launch_count() - n0 == 2
    torch.cuda.synchronize()
    assert torch.equal(l0, l1) and torch.equal(lse0, lse1) and torch.equal(g0, g1) and torch.equal(gt0, gt1)
    x = lg.double().cpu()
    ref = 0.5 * (torch.logsumexp(x, 1) + torch.logsumexp(x, 0) - 2 * x.diag()).mean()
    assert abs(float(l1) - float(ref)) < 1e-5 * max(1.0, abs(float(ref)))
    l2 = torch.zeros(1, device=DEV)
    call("lpi_clip_loss_local", n, lg, n, 1.0, 0, 0, l2, lse1[0], lse1[1], None, None, 0, s)      # value only
    assert torch.equal(l2, l0)


def test_prompt_cp_both_stacks_equal_the_per_stack_launches():
    Lyr, P, Dv, Dt, r = 9, 16, 768, 512, 4
    f = {k: torch.from_numpy(v).to(DEV) for k, v in synth.prompt_factors(Lyr, P, Dv, Dt, r=r).items()}
    d1, d2v, d2t, d3v, d3t = (f[k] for k in ("dim_1_share", "dim_2_visual", "dim_2_textual", "dim_3_visual", "dim_3_textual"))
    v0, t0 = E.prompt_cp_fwd(d1, d2v, d3v), E.prompt_cp_fwd(d1, d2t, d3t)
    n0 = _lib.launch_count()
    v1, t1 = E.prompt_cp_fwd2(d1, d2v, d2t, d3v, d3t)
    assert _lib.launch_count() - n0 == 1
    assert torch.equal(v0, v1) and torch.equal(t0, t1)
    gv, gt = rnd(Lyr, P, Dv, seed=1).to(DEV), rnd(Lyr, P, Dt, seed=2).to(DEV)
    g1 = torch.empty_like(d1)
    g2v, g3v = E.prompt_cp_bwd(d1, d2v, d3v, gv, g1, False)
    g2t, g3t = E.prompt_cp_bwd(d1, d2t, d3t, gt, g1, True)
    n0 = _lib.launch_count()
    h1, h2v, h2t, h3v, h3t = E.prompt_cp_bwd2(d1, d2v, d2t, d3v, d3t, gv, gt)
    assert _lib.launch_count() - n0 == 2
    for a, b in ((g1, h1), (g2v, h2v), (g2t, h2t), (g3v, h3v), (g3t, h3t)):
        assert torch.equal(a, b)
    # f64 reference of the shared factor's gradient
    ref = (torch.einsum("lpd,pr,dr->lr", gv.double().cpu(), d2v.double().cpu(), d3v.double().cpu())
           + torch.einsum("lpd,pr,dr->lr", gt.double().cpu(), d2t.double().cpu(), d3t.double().cpu())) / r
    assert (h1.double().cpu() - ref).abs().max() <= 1e-5 * ref.abs().max()


def test_align_loss_two_launch_form_equals_the_three_launch_form():
    Lyr, P, Dv, Dt = 9, 16, 768, 512
    vis, txt = (rnd(Lyr, P, Dv, seed=1) * 0.05).to(DEV), (rnd(Lyr, P, Dt, seed=2) * 0.05).to(DEV)
    l0, dv0, dt0 = torch.zeros(1, device=DEV), torch.zeros_like(vis), torch.zeros_like(txt)
    call("lpi_align_loss_fwd_bwd", Lyr, P, Dv, Dt, vis, txt, 0.01, 0.1, l0, dv0, dt0, stream())
    n0 = _lib.launch_count()
    l1, dv1, dt1 = E.align_loss_fwd_bwd(vis, txt, 0.01, 0.1, True)
    assert _lib.launch_count() - n0 == 2
    torch.cuda.synchronize()
    assert torch.equal(l0, l1) and torch.equal(dv0, dv1) and torch.equal(dt0, dt1)
    l2, _, _ = E.align_loss_fwd_bwd(vis, txt, 0.01, 0.1, False)
    assert torch.equal(l2, l0)


def test_transpose_pair():
    a, b = rnd(300, 512, seed=1).to(DEV), rnd(256, 384, seed=2).to(DEV)
    at, bt = torch.zeros(512, 300, device=DEV), torch.zeros(384, 256, device=DEV)
    call("lpi_transpose2", F32, 300, 512, a, 512, at, 300, 256, 384, b, 384, bt, 256, stream())
    assert torch.equal(at, a.t().contiguous()) and torch.equal(bt, b.t().contiguous())


# ------------------------------------------------------------------------------------------------ the step
def _tiny(dtype="f32"):
    from lpi_amd.engine import DualEncoder
    cfg = synth.TINY
    enc = DualEncoder(cfg, synth.clip_state_dict(cfg), dtype=dtype, device=DEV)
    img = torch.from_numpy(synth.images(4, cfg.image_resolution)).to(DEV)
    ids = torch.from_numpy(synth.token_ids(4)).to(DEV)
    fac = {k: torch.from_numpy(v).to(DEV).requires_grad_(True) for k, v in synth.prompt_factors(9, 16, cfg.vision_width, cfg.transformer_width).items()}
    return cfg, enc, img, ids, fac


def test_seeded_step_equals_the_autograd_sum_and_a_held_gradient_does_not_change():
    """train_step seeds the alignment gradient into the towers' prompt-gradient buffers (no sum kernel); the loss-graph path lets autograd add the two
    gradients.  Both give the same factor gradients (bit for bit: the same two addends per element).  And a prompt-stack gradient obtained through
    autograd (a private copy of the persistent buffer) is unchanged by the engine's next backward."""
    from lpi_amd.functional import DecomposedPromptFn, EncodeBothFn
    from lpi_amd.step import forward_loss, train_step
    cfg, enc, img, ids, fac = _tiny()
    train_step(enc, img, ids, fac, 2)
    g_step = {k: v.grad.clone() for k, v in fac.items()}
    for v in fac.values():
        v.grad = None
    losses, *_ = forward_loss(enc, img, ids, fac, 2)
    (losses["base_loss"] + losses["alignment_loss"]).backward()
    for k in fac:
        assert float((fac[k].grad - g_step[k]).abs().max()) <= 1e-6 * float(g_step[k].abs().max()) + 1e-12, k
    # a gradient of the prompt stacks taken through autograd, held across another step
    names = ("dim_1_share", "dim_2_visual", "dim_2_textual", "dim_3_visual", "dim_3_textual")
    vis, txt = DecomposedPromptFn.apply(*[fac[k] for k in names])
    vis.retain_grad(); txt.retain_grad()
    fi, ft = EncodeBothFn.apply(enc, img, ids, vis, txt, 2)
    (fi.sum() + ft.sum()).backward()
    held_v, held_t = vis.grad, txt.grad
    snap_v, snap_t = held_v.clone(), held_t.clone()
    train_step(enc, img * 1.5, ids, fac, 2)          # another batch through the same engine: its buffers are rewritten
    fi2, ft2 = EncodeBothFn.apply(enc, img * 0.5, ids, *DecomposedPromptFn.apply(*[fac[k] for k in names]), 2)
    (fi2.sum() - ft2.sum()).backward()
    torch.cuda.synchronize()
    assert torch.equal(held_v, snap_v) and torch.equal(held_t, snap_t)
    # the rows behind the prompt depth of a plain (unseeded) backward are zero again after a seeded step
    assert float(held_v[2:].abs().max()) == 0.0 and float(held_t[2:].abs().max()) == 0.0


@pytest.fixture(scope="module")
def full():
    """The benchmarked configuration: ViT-B/16, 256 pairs, bf16 (BASELINE.json configs[2])."""
    from lpi_amd.engine import DualEncoder
    cfg = synth.CONFIGS["ViT-B/16"]
    enc = DualEncoder(cfg, synth.clip_state_dict(cfg), dtype="bf16", device=DEV)
    img = torch.from_numpy(synth.images(256, cfg.image_resolution)).to(DEV)
    ids_host = synth.token_ids(256)
    return cfg, enc, img, ids_host


LAUNCHES_PER_STEP_MAX = 220      # round 3: 259 (profiles/r04_*_step_sequence.txt lists them); the tail's launches are paired / fused since (224); round 6: + 6 — the
                                 # vision tower's last block runs without K and V as three launches each way (csrc/attn_stream.hip) instead of riding in the towers' pair
                                 # launches — and - 13: the few-row GEMMs are ONE launch each (csrc/gemm_rows.hip) instead of split-K partial + reduction: 217


def test_steady_state_launch_count_and_no_foreign_kernel_in_the_step(full):
    """The benchmarked step issues a fixed number of library launches — the towers' tail kernels paired, the loss / CP / alignment kernels fused — and
    NOTHING else: every device kernel the profiler sees inside a steady-state step is one of the library's (no ATen kernel: the alignment gradient is
    seeded into the towers' buffers instead of being added by autograd)."""
    from lpi_amd.optim import FlatSGD, flatten
    from lpi_amd.step import train_step
    cfg, enc, img, ids_host = full
    fac = {k: torch.from_numpy(v).to(DEV).requires_grad_(True) for k, v in synth.prompt_factors(9, 16, cfg.vision_width, cfg.transformer_width).items()}
    flat, flat_grad, views = flatten(fac)
    opt = FlatSGD(fac, lr=0.05, momentum=0.9, weight_decay=2e-4, flat=flat, flat_grad=flat_grad, grad_views=views)
    pk = E.PackedIds(np.ascontiguousarray(E.trim_token_ids(ids_host))).to(DEV)

    def step():
        train_step(enc, img, pk, fac, 3, flat_grad=flat_grad, grad_views=views)
        opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    n0 = _lib.launch_count()
    step()
    n1 = _lib.launch_count()
    step()
    n2 = _lib.launch_count()
    print(f"\n    launches per steady-state step: {n1 - n0}")
    assert n1 - n0 == n2 - n1
    assert n1 - n0 <= LAUNCHES_PER_STEP_MAX, n1 - n0
    from torch.profiler import ProfilerActivity, profile
    dev_events = []
    for attempt in range(8):      # the tracer now and then hands back a fraction of a window's records (seen in rounds 5 and 6): profile another step then
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            step()
            torch.cuda.synchronize()
        dev_events = [e.name for e in prof.events() if e.device_type.name != "CPU"]
        if len(dev_events) >= n1 - n0:
            break
    foreign = [n for n in dev_events if "anonymous namespace" not in n and "_GLOBAL__N_" not in n and "lpi" not in n.lower() and "StatFin" not in n
               and "Memcpy" not in n and "Memset" not in n]
    assert len(dev_events) >= n1 - n0 and foreign == [], foreign


def test_lockstep_towers_realign_when_one_tower_takes_the_statistics_pass(full):
    """One tower at LPI_ROWSTATS = 0 (a statistics pass per LayerNorm) and the other at 2 (statistics from the GEMM epilogues + finalize launches): their
    request streams differ in the CONDITIONAL requests only, and run_lockstep must keep pairing the GEMMs (engine.run_lockstep: an optional request is
    issued alone and only its tower advances).  Same bits as the sequential towers; about as few launches as the aligned case."""
    cfg, enc, img, ids_host = full
    ids = torch.from_numpy(ids_host).to(DEV)
    f = {k: torch.from_numpy(v).to(DEV) for k, v in synth.prompt_factors(9, 16, cfg.vision_width, cfg.transformer_width).items()}
    vis, txt = E.prompt_cp_fwd2(f["dim_1_share"], f["dim_2_visual"], f["dim_2_textual"], f["dim_3_visual"], f["dim_3_textual"])

    def both(lock):
        n0 = _lib.launch_count()
        if lock:
            (fi, _), (ft, _) = enc.encode_both(img, ids, vis, txt, 3, train=True)
        else:
            fi = enc.encode_image(img, vis, 3, train=True)
            ft = enc.encode_text(ids, txt, 3, train=True)
        torch.cuda.synchronize()
        return fi.clone(), ft.clone(), _lib.launch_count() - n0
    a_i, a_t, n_aligned = both(True)
    try:
        enc.txt.rowstats = 0
        b_i, b_t, n_mixed = both(True)
        c_i, c_t, n_seq = both(False)
    finally:
        enc.txt.rowstats = enc.opt.rowstats
    print(f"\n    forward launches: aligned {n_aligned}, one tower on the statistics pass {n_mixed}, sequential towers {n_seq}")
    assert torch.equal(b_i, c_i) and torch.equal(b_t, c_t)          # lock step == sequential, bit for bit
    assert torch.equal(a_i, b_i)                                      # the vision tower is untouched by the text tower's switch
    # per block the text tower now issues two statistics passes of its own (its finalizes were paired with the vision tower's) and its two residual GEMMs
    # have another epilogue kind than the vision tower's (no row statistics), so those two pairs split: ~4 more launches per block — but every other GEMM,
    # attention and tail pair is still ONE launch (shifted by one request they would all run ungrouped: the sequential towers' count)
    assert n_mixed <= n_aligned + 4 * cfg.transformer_layers + 2 and n_mixed < n_seq - 30, (n_aligned, n_mixed, n_seq)


def test_flat_sgd_refuses_reseated_parameters():
    from lpi_amd.optim import FlatSGD
    p = [torch.nn.Parameter(torch.randn(5, 3, device=DEV)), torch.nn.Parameter(torch.randn(7, device=DEV))]
    opt = FlatSGD(p, lr=0.1, momentum=0.9)
    for q in p:
        q.grad = torch.ones_like(q)
    opt.step()
    p[1].data = p[1].data.clone()            # what module.to(...) / load_state_dict(assign=True) does
    with pytest.raises(RuntimeError, match="no longer a view"):
        opt.step()


# ------------------------------------------------------------------------------------------------ accuracy envelope of the one-sweep statistics
def test_single_sweep_statistics_accuracy_envelope():
    """var = E[x^2] - mean^2 in f32 (lpi_ln_stats_finalize from the GEMM epilogue's slot sums; row_stats_1sweep in the front ends) against two-pass f64
    statistics of the same fp16 rows: rows of a CLIP residual stream (|mean| below the deviation), rows with 'massive activation' channels (a few
    channels hundreds of deviations out: large variance, small relative error), and rows with |mean| = 10 / 30 deviations, where the form loses
    digits as (mean / std)^2 * 1e-7 — the envelope that LPI_ROWSTATS = 0 (the two-pass statistics pass) removes."""
    d, M = 768, 512
    s = stream()
    cases = {}
    base = rnd(M, d, seed=1)
    cases["stream"] = (base * 0.7 + 0.1, 2e-5)
    massive = base.clone()
    massive[:, 5] += 300.0
    massive[:, 400] -= 180.0
    cases["massive activations"] = (massive, 2e-5)
    cases["mean = 10 std"] = (base + 10.0, 2e-4)
    cases["mean = 30 std"] = (base + 30.0, 2e-3)
    for name, (x, bar) in cases.items():
        x16 = x.half()
        xs = x16.double()
        part = torch.stack([torch.stack([xs[:, j * 128:(j + 1) * 128].sum(1), (xs[:, j * 128:(j + 1) * 128] ** 2).sum(1)]) for j in range(d // 128)])
        part = part.reshape(2 * (d // 128), M).float().to(DEV).contiguous()          # exact slot sums, rounded to f32 once (the epilogue's are f32 sums)
        mean, rstd = torch.zeros(M, device=DEV), torch.zeros(M, device=DEV)
        call("lpi_ln_stats_finalize", M, d, part, M, 1e-5, mean, rstd, s)
        ref_rstd = 1.0 / (xs.var(1, unbiased=False) + 1e-5).sqrt()
        err = float(((rstd.double().cpu() / ref_rstd) - 1).abs().max())
        assert err <= bar, (name, err)
        assert float((mean.double().cpu() - xs.mean(1)).abs().max()) <= 1e-5 * max(1.0, float(xs.mean(1).abs().max())), name
        # the same rows through the statistics pass (two sweeps): exact to f32 round-off whatever the mean
        m2, r2 = torch.zeros(M, device=DEV), torch.zeros(M, device=DEV)
        call("lpi_layernorm_fwd", BF16, F16, M, d, x16.to(DEV), d, None, None, None, 0, m2, r2, s)
        assert float(((r2.double().cpu() / ref_rstd) - 1).abs().max()) <= 2e-6, name


def test_bench_multi_rank_line_on_one_gpu_reports_the_observed_world_size():
    """`bench.py --gpus 2 --share-gpu` (two ranks on this GPU, gloo group, host-staged messages): the ranks are started as a CHILD torch.distributed.run
    before anything touches the GPU in the launcher, rank 0's JSON line carries the world size the process group observed and the whole-job value —
    the multi-rank path of the bench is exercised on a one-GPU box (the 8-GPU run is the driver's)."""
    import json
    import os
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for extra in ([], ["--local-loss", "--gather-with-grad"]):
        p = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--share-gpu", "--batch", "16", "--steps", "2", "--warmup", "1",
                            "--no-cpu-baseline", "--no-roofline", "--no-extras"] + extra, capture_output=True, text=True, env=env, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln][-1]
        j = json.loads(line)
        assert j["n_gpus"] == 2 and j["collectives"]["observed_world_size"] == 2 and j["collectives"]["backend"] == "gloo"
        assert j["config"]["global_batch"] == 32 and j["value"] > 0 and j["scaling"] == "weak"
        assert ("local_loss=True" in j["config"]["dp_mode"]) == bool(extra)


@pytest.mark.parametrize("tm,N,K", [(3, 512, 128), (43, 1536, 512), (213, 768, 768), (100, 768, 3072), (30, 3072, 256), (90, 768, 2304)])
def test_gemm_half_width_staging_epilogue_equals_the_generic_one(tm, N, K):
    """EPI_PLAIN16 (gemm256p.hip): a store-only GEMM with no bias, alpha = 1 and a 2-byte output rounds its accumulators BEFORE the LDS staging (half the
    staging bytes, double-buffered passes, 16-byte row stores).  Bit for bit the generic epilogue (tuning key 14 = 1) — persistent tiles, the hybrid
    half-tile round, a grouped launch — and the generic one is held against f64 by test_kernels_gpu.py."""
    M = tm * 256
    a = rnd(M, K, seed=tm).bfloat16().to(DEV)
    a[5] = 0                                        # a row of exact zeros (the sign of a stored zero)
    b = (rnd(N, K, seed=tm + 1) * 0.05).bfloat16().to(DEV)
    outs = []
    try:
        for key in (0, 1):
            call("lpi_set_tuning", 14, key)
            c = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=DEV)
            E.gemm(BF16, a, b, c, M, N, K)
            torch.cuda.synchronize()
            outs.append(c)
        # grouped with a second problem
        a2 = rnd(1024, 512, seed=9).bfloat16().to(DEV)
        b2 = (rnd(512, 512, seed=10) * 0.05).bfloat16().to(DEV)
        grp = []
        for key in (0, 1):
            call("lpi_set_tuning", 14, key)
            c, c2 = torch.zeros(M, N, dtype=torch.bfloat16, device=DEV), torch.zeros(1024, 512, dtype=torch.bfloat16, device=DEV)
            _lib.gemm_grouped(BF16, BF16, E.EPI_NONE, 1.0, [dict(M=M, N=N, K=K, a=a, b=b, c=c), dict(M=1024, N=512, K=512, a=a2, b=b2, c=c2)], stream())
            torch.cuda.synchronize()
            grp.append((c, c2))
    finally:
        call("lpi_set_tuning", 14, 0)
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    assert torch.equal(grp[0][0].view(torch.int16), grp[1][0].view(torch.int16)) and torch.equal(grp[0][1].view(torch.int16), grp[1][1].view(torch.int16))
    assert torch.equal(grp[0][0].view(torch.int16), outs[0].view(torch.int16))
    ref = a.double().cpu() @ b.double().cpu().t()
    assert float((outs[0].double().cpu() - ref).abs().max()) <= 2e-2 * float(ref.abs().max())


def test_device_kmeans_equals_the_reference_clustering_and_the_oracle(golden):
    """lpi_amd.kmeans.kmeans_fit (HIP passes over the features, scikit-learn's seeding and convergence logic on the host) on the synthetic features of the
    fixture: the centres the IMPORTED reference's clustering() found (sprompt.py:370-397, tests/golden/kmeans.npz) to 1e-5, the oracle's labels exactly —
    and through the plugin's SPrompts.clustering, which normalises with lpi_l2norm_fwd."""
    from lpi_amd.kmeans import kmeans_fit
    from oracle import lpi_oracle as O
    g = golden("kmeans")
    n, dim = (int(x) for x in g["shape"])
    feats = synth.clustering_features(n, dim)
    n0 = _lib.launch_count()
    for name, f in zip(("visual", "textual"), feats):
        x = f / np.linalg.norm(f, axis=-1, keepdims=True)
        centers, labels, iters = kmeans_fit(torch.from_numpy(x).to(DEV), 5, random_state=0)
        oc, ol, oi = O.kmeans_fit(x)
        assert np.abs(centers.cpu().numpy() - g["centers_" + name]).max() < 1e-5, name
        assert np.array_equal(labels.cpu().numpy(), ol) and iters == oi, name
    assert _lib.launch_count() - n0 >= 2 * (1 + 5 + 2)
    with pytest.raises(_lib.LpiError):
        kmeans_fit(torch.from_numpy(feats[0]), 5)              # host tensor: no CPU fallback
    # ... and as the plugin calls it
    from lpi_amd.retrieval.methods.sprompt import SPrompts
    fv, ft = (torch.from_numpy(x).to(DEV) for x in feats)

    class Net:
        def extract_vector(self, idx):
            return fv[idx]

        def extract_textual_vector(self, idx):
            return ft[torch.as_tensor(idx, device=DEV)]

    class Loader:
        def __iter__(self):
            for i in range(0, n, 128):
                idx = torch.arange(i, min(n, i + 128))
                yield idx, idx, None, None

    sp = object.__new__(SPrompts)
    sp._network, sp._device, sp.all_keys, sp.textual_all_keys, sp.args = Net(), torch.device(DEV), [], [], {}
    sp.clustering(Loader())
    assert np.abs(sp.all_keys[0].cpu().numpy() - g["centers_visual"]).max() < 1e-5
    assert np.abs(sp.textual_all_keys[0].cpu().numpy() - g["centers_textual"]).max() < 1e-5

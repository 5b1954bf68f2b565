"""BASELINE.json configs[4] at its stated per-GPU size: ViT-L/14 (24 x 1024-wide vision layers, 12 x 768-wide text layers, patch 14 -> 257 + 16
tokens, embed 768), prompt_depth 12, CP rank 8, 512 pairs per GPU, bf16 — through size-independent properties, as tests/test_fullsize_gpu.py does
for ViT-B/16 (the oracle needs minutes per sample at this size): unit-norm features, the loss recomputed on the CPU from the features, the packed
text batch against the trimmed one, finite gradients of every factor; and, in the f32 parity mode at 128 pairs (the f32 workspace of 512 pairs
does not fit beside the bf16 one), central finite differences of the total loss against the hand-written backward.
The architecture itself is pinned against the oracle at batch 2-3 in tests/test_fullsize_gpu.py."""
import gc

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from lpi_amd import synth  # noqa: E402
from lpi_amd.engine import DualEncoder, PackedIds, trim_token_ids  # noqa: E402
from lpi_amd.step import forward_loss, train_step  # noqa: E402

DEV = "cuda:0"
CFG = synth.VIT_L14
B, DEPTH, R = 512, 12, 8


def factors(requires_grad=True):
    return {k: torch.from_numpy(v).to(DEV).requires_grad_(requires_grad)
            for k, v in synth.prompt_factors(12, 16, CFG.vision_width, CFG.transformer_width, r=R).items()}


@pytest.fixture(scope="module")
def weights():
    return synth.clip_state_dict(CFG)


def _mem(tag):
    print(f"[HBM] {tag}: allocated {torch.cuda.memory_allocated() / 2**30:.1f} GiB, reserved {torch.cuda.memory_reserved() / 2**30:.1f} GiB")


def test_vit_l14_512_pairs_bf16_step_properties_and_packed_text(weights):
    gc.collect()
    torch.cuda.empty_cache()
    _mem("before the 512-pair step")
    img = torch.from_numpy(synth.images(B, CFG.image_resolution)).to(DEV)
    ids_h = synth.token_ids(B)
    ids_t = torch.from_numpy(np.ascontiguousarray(trim_token_ids(ids_h))).to(DEV)
    pk = PackedIds(ids_h).to(DEV)
    enc = DualEncoder(CFG, weights, dtype="bf16", device=DEV)
    res = {}
    for tag, ids in (("trim", ids_t), ("pack", pk)):
        fac = factors()
        out = train_step(enc, img, ids, fac, DEPTH)
        torch.cuda.synchronize()
        res[tag] = ({k: v.clone() for k, v in out.items()}, {k: fac[k].grad.clone() for k in synth.PROMPT_NAMES})
    out, grads = res["pack"]
    i_f, t_f = out["img_f"].double().cpu(), out["txt_f"].double().cpu()
    assert i_f.shape == (B, 768) and t_f.shape == (B, 768)
    assert torch.allclose(i_f.norm(dim=1), torch.ones(B, dtype=torch.float64), atol=1e-5)
    assert torch.allclose(t_f.norm(dim=1), torch.ones(B, dtype=torch.float64), atol=1e-5)
    lg = enc.logit_scale_exp * i_f @ t_f.t()
    lab = torch.arange(B)
    ce = (torch.nn.functional.cross_entropy(lg, lab) + torch.nn.functional.cross_entropy(lg.t(), lab)) / 2
    assert abs(float(out["base_loss"]) - float(ce)) < 2e-5 * max(1.0, float(ce))          # the loss kernels are f32 whatever the towers' mode
    for k in synth.PROMPT_NAMES:
        g = grads[k]
        assert g is not None and bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0, k
    # packed (one row per live token) against trimmed (cut at the longest caption): the same arithmetic per live row
    for k in ("img_f", "txt_f"):
        assert float((res["trim"][0][k] - res["pack"][0][k]).abs().max()) < 1e-3, k
    assert abs(float(res["trim"][0]["base_loss"]) - float(res["pack"][0]["base_loss"])) < 1e-3
    for k in synth.PROMPT_NAMES:
        a, b = res["pack"][1][k].double(), res["trim"][1][k].double()
        assert float((a - b).abs().max()) <= 2e-2 * float(b.abs().max()), k
    print(f"ViT-L/14, 512 pairs, depth 12, r 8, bf16: base loss {float(out['base_loss']):.4f} (ln 512 = {np.log(512):.4f}); packed text {pk.rows / B:.1f} rows per caption")
    del enc, fac
    gc.collect()                # the engine holds reference cycles (towers <-> workspaces): collect them before handing the HBM back
    torch.cuda.empty_cache()


def test_vit_l14_finite_difference_of_total_loss_f32(weights):
    """d(base + alignment loss)/d(factor entry) from the hand-written backward against central differences: ViT-L/14, depth 12, r 8, f32 mode."""
    gc.collect()
    torch.cuda.empty_cache()
    _mem("before the finite-difference test")
    Bf = 128
    img = torch.from_numpy(synth.images(Bf, CFG.image_resolution)).to(DEV)
    ids = PackedIds(synth.token_ids(Bf)).to(DEV)
    enc = DualEncoder(CFG, weights, dtype="f32", device=DEV)
    fac = factors()
    train_step(enc, img, ids, fac, DEPTH)
    grads = {k: fac[k].grad.clone() for k in synth.PROMPT_NAMES}

    def total(f):
        with torch.no_grad():
            losses, *_ = forward_loss(enc, img, ids, f, DEPTH)
        return float(losses["base_loss"].double() + losses["alignment_loss"].double())

    for name in ("dim_1_share", "dim_2_visual", "dim_3_textual"):
        # the entry with the largest gradient of the factor: the f32 loss carries ~1e-6 of round-off, i.e. ~3e-5 in the difference quotient
        flat_i = int(grads[name].abs().argmax())
        idx = (flat_i // grads[name].shape[1], flat_i % grads[name].shape[1])
        eps = 2e-2
        f = factors(False)
        f[name][idx] += eps
        lp = total(f)
        f[name][idx] -= 2 * eps
        lm = total(f)
        fd = (lp - lm) / (2 * eps)
        g = float(grads[name][idx])
        print(f"ViT-L/14 f32 finite difference {name}{idx}: {fd:.6f} vs backward {g:.6f}")
        assert abs(g) > 2e-4 and abs(fd - g) <= 0.05 * abs(g) + 2e-5, (name, idx, fd, g)
    del enc, fac
    gc.collect()                # the engine holds reference cycles (towers <-> workspaces): collect them before handing the HBM back
    torch.cuda.empty_cache()

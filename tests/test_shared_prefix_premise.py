"""The premise of the text tower's shared-prefix layout (include/lpi_hip.h: lpi_attn_fwd_shared; engine.PackedIds(shared=17)), stated on the ORACLE — the CPU
restatement of the reference's own arithmetic (oracle/lpi_oracle.py: prompt_learner.py:128-163 splice, model.py:189-207 blocks, model.py:347-353 causal mask):

  in the training forward (prompts broadcast over the batch, slinet.py:119-130) positions 0 .. n_ctx of the text tower's residual stream hold the SAME rows for
  every sample at the input of every block, and no caption token can change them.

The HIP layout stores those rows once; this test pins WHY that is exact, independent of any kernel."""
import numpy as np
import torch

from lpi_amd import synth
from oracle import lpi_oracle as O


def _streams(oracle, ids, txt_prompts, depth):
    """The residual stream at the input of every text block (and the tower's output)."""
    seen = []
    inner = O.res_block

    def spy(x, W, pre, heads, causal):
        if pre.startswith("transformer."):
            seen.append(x.detach().clone())
        y = inner(x, W, pre, heads, causal)
        if pre.startswith("transformer.") and pre.endswith(f"resblocks.{oracle.cfg.transformer_layers - 1}."):
            seen.append(y.detach().clone())
        return y
    O.res_block = spy
    try:
        emb = oracle.text_embed(torch.from_numpy(ids), txt_prompts[0])
        oracle.encode_text(emb, torch.from_numpy(ids), txt_prompts.unsqueeze(0).expand(ids.shape[0], -1, -1, -1), depth)
    finally:
        O.res_block = inner
    return seen


def test_first_positions_are_the_same_rows_for_every_sample_in_every_block():
    cfg = synth.TINY
    orc = O.Oracle(cfg, synth.clip_state_dict(cfg), torch.float64)
    fac = {k: torch.from_numpy(v).double() for k, v in synth.prompt_factors(9, 16, cfg.vision_width, cfg.transformer_width).items()}
    _, txt = O.decomposed_prompt(fac)
    ids = synth.token_ids(6)
    n = 17
    a = _streams(orc, ids, txt, depth=2)
    assert len(a) == cfg.transformer_layers + 1
    for layer, x in enumerate(a):
        for b in range(1, ids.shape[0]):
            assert torch.equal(x[b, :n], x[0, :n]), (layer, b)               # identical rows: the same operations on the same numbers
        assert float((x[1:, n:] - x[:1, n:]).abs().max()) > 1e-3               # ... while the captions' own rows do differ
    # no caption token reaches them: other captions (other lengths, other tokens), the same rows
    other = synth.token_ids(6, seed=synth.TOKEN_SEED + 99)
    assert not np.array_equal(other, ids)
    b_ = _streams(orc, other, txt, depth=2)
    for layer, (x, y) in enumerate(zip(a, b_)):
        assert torch.equal(x[0, :n], y[0, :n]), layer
    # per-sample prompt stacks (inference, slinet.py:215) break the premise: the layout is refused there (tests/test_shared_prefix_gpu.py)
    emb = orc.text_embed(torch.from_numpy(ids), txt[0])
    stacks = torch.stack([txt * (1.0 + 0.1 * b) for b in range(ids.shape[0])])
    seen = []
    inner = O.res_block

    def spy(x, W, pre, heads, causal):
        if pre.startswith("transformer.resblocks.1."):
            seen.append(x.detach().clone())
        return inner(x, W, pre, heads, causal)
    O.res_block = spy
    try:
        orc.encode_text(emb, torch.from_numpy(ids), stacks, 2)
    finally:
        O.res_block = inner
    assert float((seen[0][1, 1:n] - seen[0][0, 1:n]).abs().max()) > 1e-6

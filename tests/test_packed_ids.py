"""engine.PackedIds on the host: the plain packed layout and the shared-prefix layout of the training forward (include/lpi_hip.h: lpi_attn_fwd_shared)."""
import numpy as np
import pytest

from lpi_amd import synth
from lpi_amd.engine import PackedIds


def test_plain_and_shared_row_layout():
    ids = synth.token_ids(6)
    lens = ids.argmax(-1) + 1
    p = PackedIds(ids)
    assert p.shared == 0 and p.rows == lens.sum() and p.shape == (6, lens.max())
    assert np.array_equal(p.row_start.numpy(), np.concatenate([[0], np.cumsum(lens)]))
    assert np.array_equal(p.pool_rows.numpy(), np.cumsum(lens) - 1) and np.array_equal(p.eot.numpy(), lens - 1)
    s = PackedIds(ids, shared=17)
    own = lens - 17
    assert s.shared == 17 and s.rows == 17 + own.sum() and s.shape == p.shape
    assert np.array_equal(s.row_start.numpy(), 17 + np.concatenate([[0], np.cumsum(own)]))
    assert np.array_equal(s.pool_rows.numpy(), 17 + np.cumsum(own) - 1)          # the EOT's absolute row ...
    assert np.array_equal(s.eot.numpy(), lens - 1)                               # ... and its POSITION in the caption (unchanged)
    assert np.array_equal(s.ids.numpy(), p.ids.numpy())                          # the id matrix stays [B, longest]
    # sub-batches (data-parallel shards, micro-batches) keep the layout
    sub = s[2:5]
    assert sub.shared == 17 and sub.rows == 17 + own[2:5].sum() and int(sub.row_start[0]) == 17
    with pytest.raises(TypeError):
        s[::2]


def test_shared_layout_validates_its_premises():
    ids = synth.token_ids(3)
    short = ids.copy()
    short[0, 16], short[0, 17:] = synth.EOT, 0              # EOT inside the context slots: nothing behind the shared positions
    with pytest.raises(ValueError, match="continue behind"):
        PackedIds(short, 17)
    other = ids.copy()
    other[1, 0] -= 1
    with pytest.raises(ValueError, match="same token"):
        PackedIds(other, 17)
    PackedIds(other)                                         # the plain layout has no such premise

"""CU-partitioned lanes: HIP streams restricted to disjoint subsets of the MI355X's 256 CUs (8 XCDs x 32).

Why: the 256x256 GEMM owns a whole CU per workgroup (all its LDS and registers), so two streams without masks never co-schedule
and the chip runs every kernel in lockstep — all CUs stream HBM together (epilogues, LayerNorm, attention staging), then all run the
matrix cores together.  With the chip split in two and half the batch per lane, one lane's HBM-bound phases overlap the other
lane's MFMA-bound phases.  The reference runs a single stream (methods/sprompt.py:297-311); the batch split changes only the
summation order of the batch-summed prompt gradients.

Mask layout (HIP: bit i = CU i; the driver deals consecutive bits round-robin over the XCDs, see tools/cu_mask_probe.py):
  'xcd'  : lane k owns the XCDs {x : x % n == k}  -> disjoint L2s
  'half' : lane k owns CUs [k*32/n, (k+1)*32/n) of EVERY XCD -> both lanes share all eight L2s
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib


def cu_masks(n_lanes: int, mode: str = "xcd", n_cu: int = 256, n_xcd: int = 8):
    """-> list of n_lanes masks, each a list of 32-bit words (bit i of the concatenation = CU i)."""
    per = n_cu // n_xcd
    words = (n_cu + 31) // 32
    out = []
    for k in range(n_lanes):
        bits = 0
        for i in range(n_cu):
            xcd, idx = i % n_xcd, i // n_xcd          # consecutive mask bits go to consecutive XCDs
            if mode == "xcd":
                on = xcd % n_lanes == k
            elif mode == "half":
                on = idx * n_lanes // per == k
            elif mode == "block":                     # contiguous bit ranges (for probing the layout)
                on = i * n_lanes // n_cu == k
            else:
                raise ValueError(mode)
            if on:
                bits |= 1 << i
        out.append([(bits >> (32 * w)) & 0xFFFFFFFF for w in range(words)])
    return out


class MaskedStream:
    """Owns a CU-masked HIP stream and exposes it as a torch stream."""

    def __init__(self, mask_words, device):
        _lib.load()
        arr = (ctypes.c_uint32 * len(mask_words))(*mask_words)
        h = ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.call("lpi_stream_create_cu_mask", ctypes.addressof(arr), len(mask_words), ctypes.addressof(h))
        self.handle = h.value
        self.stream = torch.cuda.ExternalStream(self.handle, device=device)

    def close(self):
        if self.handle:
            _lib.call("lpi_stream_destroy", self.handle)
            self.handle = None


def make_lane_streams(n_lanes: int, mode: str, device):
    n_cu = _lib.load().lpi_device_cu_count()
    if n_cu <= 0:
        raise _lib.LpiError("cannot query the CU count")
    return [MaskedStream(m, device) for m in cu_masks(n_lanes, mode, n_cu)]

#!/usr/bin/env python3
"""Where do the workgroups of a CU-masked stream land?  Prints, per mask layout and lane, the XCDs (HW_REG_XCC_ID) and the number of
distinct CUs (XCC, SE, SH, CU fields of HW_REG_HW_ID) that a one-workgroup-per-CU grid reached."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lpi_amd import _lib, lanes  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
_lib.load()
print("CUs:", _lib.load().lpi_device_cu_count())


def probe(stream, blocks=1024):
    out = torch.zeros(2 * blocks, dtype=torch.int32, device=dev)
    _lib.call("lpi_probe_placement", blocks, 512, 128 * 1024, 200, out, stream.cuda_stream)
    stream.synchronize()
    o = out.cpu().numpy().astype("uint32").reshape(-1, 2)
    xcc = o[:, 0] & 0xF
    hw = o[:, 1]
    cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 0x1, (hw >> 13) & 0x7
    cus = {(int(a), int(b), int(c), int(d)) for a, b, c, d in zip(xcc, se, sh, cu)}
    per = {}
    for a in xcc:
        per[int(a)] = per.get(int(a), 0) + 1
    return sorted(per.items()), len(cus)


print("unmasked:", probe(torch.cuda.current_stream()))
for mode in ("xcd", "half", "block"):
    for n in (2, 4):
        ss = lanes.make_lane_streams(n, mode, dev)
        for k, ms in enumerate(ss):
            print(f"mode={mode} lanes={n} lane={k}: workgroups per XCD, distinct CUs =", probe(ms.stream))
        for ms in ss:
            ms.close()

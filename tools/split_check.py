import os, sys, torch
sys.path.insert(0, '/root/repo')
from lpi_amd import synth, engine as E
from lpi_amd.engine import DualEncoder
cfg = synth.CONFIGS["ViT-B/16"]
enc = DualEncoder(cfg, synth.clip_state_dict(cfg), dtype="bf16", device="cuda:0")
img = torch.from_numpy(synth.images(256, 224)).to("cuda:0")
ids = torch.from_numpy(synth.token_ids(256)).to("cuda:0")
f = {k: torch.from_numpy(v).to("cuda:0") for k, v in synth.prompt_factors(9, 16, 768, 512).items()}
vis, txt = E.prompt_cp_fwd2(f["dim_1_share"], f["dim_2_visual"], f["dim_2_textual"], f["dim_3_visual"], f["dim_3_textual"])
outs = []
for sp in (0, 128):
    E.MLP_SPLIT = sp
    enc.vis._ws.clear()
    (fi, _), (ft, _) = enc.encode_both(img, ids, vis, txt, 3, train=True)
    torch.cuda.synchronize()
    outs.append((fi.clone(), ft.clone()))
print("equal:", torch.equal(outs[0][0], outs[1][0]), torch.equal(outs[0][1], outs[1][1]), float(outs[0][0].abs().max()))

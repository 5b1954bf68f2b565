import gc, sys, types, torch
sys.path.insert(0, "/root/repo")
from lpi_amd import synth
from lpi_amd.engine import DualEncoder, PackedIds
from lpi_amd.step import train_step
cfg = synth.TINY
sd = synth.clip_state_dict(cfg)
enc = DualEncoder(cfg, sd, dtype="bf16", device="cuda:0")
fac = {k: torch.from_numpy(v).to("cuda:0").requires_grad_(True) for k, v in synth.prompt_factors(9, 16, cfg.vision_width, cfg.transformer_width).items()}
img = torch.from_numpy(synth.images(4, cfg.image_resolution)).to("cuda:0")
ids = PackedIds(synth.token_ids(4)).to("cuda:0")
out = train_step(enc, img, ids, fac, 2)
torch.cuda.synchronize()
del enc, fac, img, ids, out
gc.collect()
print("allocated after del + gc:", round(torch.cuda.memory_allocated() / 2**20, 1), "MiB")
objs = [o for o in gc.get_objects() if type(o).__name__ == "DualEncoder"]
print("live DualEncoder objects:", len(objs))
def show(o, depth=0, seen=None):
    seen = seen or set()
    if depth > 4 or id(o) in seen:
        return
    seen.add(id(o))
    for r in gc.get_referrers(o):
        if r is objs or isinstance(r, types.FrameType) or id(r) in seen:
            continue
        desc = type(r).__name__
        if isinstance(r, dict):
            desc += " keys=" + str(list(r.keys())[:6])
        elif isinstance(r, (tuple, list)):
            desc += f" len={len(r)}"
        elif isinstance(r, types.FunctionType):
            desc += " " + r.__qualname__
        print("  " * depth + "<- " + desc[:160])
        show(r, depth + 1, seen)
for o in objs[:1]:
    show(o)

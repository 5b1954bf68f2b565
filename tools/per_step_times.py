import sys, json, numpy as np
sys.path.insert(0, '.')
import bench, torch, types
a = types.SimpleNamespace(gpus=1, steps=40, warmup=5, dtype='bf16', batch=256, depth=3, model='ViT-B/16', rank=4, prompt_layers=9, fwd_only=False, overlap=False, no_lockstep=False, no_text_pack=False, no_text_trim=False, vision_lanes=1, text_lanes=1)
dev = torch.device('cuda:0'); torch.cuda.set_device(0)
wl = bench.Workload(a, dev, 0, 'bf16', False, None)
for rep in range(3):
    el, per = wl.run(40, 5, torch.cuda.synchronize)
    print(rep, round(1e3*el/40,3), 'median', round(float(np.median(per)),3), 'min', round(min(per),3), 'max', round(max(per),3), [round(p,2) for p in per[:12]])

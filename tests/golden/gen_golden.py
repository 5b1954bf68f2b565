#!/usr/bin/env python3
"""Generate golden fixtures by IMPORTING the reference (Kelvin-ywc/LPI, retrieval/) in this container.

Runs only where /root/reference exists (the build container).  Nothing of the reference travels:
the outputs are small ``.npz`` files of inputs/expected outputs under ``tests/golden/``.

The reference cannot run on CPU or offline unmodified.  The harness applies
exactly the shim list of SURVEY.md section 8(c):

  module stubs : torchvision(+transforms,+datasets), ftfy, loguru, models.sinet
  runtime shims: Tensor.cuda -> identity, torch.cuda.current_device -> 0, torch.cuda.device_count -> 1,
                 models.slinet.load_clip_to_cpu -> CLIP(...).float().eval() with synthetic weights
                 (skips the download and the fp16 convert_weights: F3), cwd = retrieval/ (./MID/...)

A third family (round 5), ``*_fp16``: the reference in ITS OWN arithmetic type — ``build_model``'s ``convert_weights`` (model.py:394-415, 522) applied, so
``SliNet.dtype`` is fp16 and images / prompts are cast to it (slinet.py:30, 117-128), run by torch's CPU fp16 kernels.  It pins the ``compute_dtype='f16'``
mode of the build to the reference's fp16 behaviour (within fp16 rounding: accumulation orders differ), next to the f32 families that pin the parity mode.

Two golden families (F1):
  * ``*_d1``          true oracle: the shipped code, whose deep-prompt guard is dead => depth 1.
  * ``*_d3_patched``  patched oracle: ResidualAttentionBlock.forward re-stated in this harness with the
                      intended guard ``0 < layer_id < depth`` (model.py:190-193), depth = 3.

Usage:  python tests/golden/gen_golden.py [--only tiny|vitb16]
"""
import argparse
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/retrieval"
sys.path.insert(0, REPO)

from lpi_amd import synth  # noqa: E402

torch.set_num_threads(8)


# ----------------------------------------------------------------------------- shims
def install_shims():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _T:  # transform placeholder
        def __init__(self, *a, **k):
            pass

        def __call__(self, x):
            return x

    tv = stub("torchvision")
    names = ["Compose", "Resize", "CenterCrop", "ToTensor", "Normalize", "RandomResizedCrop",
             "RandomHorizontalFlip", "ColorJitter"]
    tr = stub("torchvision.transforms", **{n: _T for n in names})
    tr.InterpolationMode = types.SimpleNamespace(BICUBIC=3)
    tv.transforms = tr
    tv.datasets = stub("torchvision.datasets")
    stub("ftfy", fix_text=lambda s: s)

    class _Logger:
        def add(self, *a, **k):
            return 0

        def info(self, *a, **k):
            pass

    stub("loguru", logger=_Logger())
    stub("models.sinet", SiNet=object)

    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.cuda.current_device = lambda: 0
    torch.cuda.device_count = lambda: 1

    sys.path.insert(0, REF)
    os.chdir(REF)  # slinet.py:171 reads ./MID/task_sim_matrix.txt relative to cwd


def ref_args(cfg: synth.ClipConfig):
    with open(os.path.join(REF, "configs/lpi/coco_lpi.json")) as f:
        args = json.load(f)
    args["device"] = [torch.device("cpu")]
    args["visual_dim"] = cfg.vision_width
    args["textual_dim"] = cfg.transformer_width
    return args


def build_slinet(cfg: synth.ClipConfig, fp16: bool = False):
    import models.slinet as slinet
    from models.clip.model import CLIP, convert_weights

    def load_clip_to_cpu(_args):
        sd = {k: torch.from_numpy(v) for k, v in synth.clip_state_dict(cfg).items()}
        if fp16:      # build_model's own sequence (model.py:516-523): CLIP(...), convert_weights, load_state_dict, eval
            model = CLIP(*cfg.as_clip_args())
            convert_weights(model)
            model.load_state_dict(sd)
            return model.eval()
        model = CLIP(*cfg.as_clip_args()).float().eval()
        model.load_state_dict(sd)
        return model

    slinet.load_clip_to_cpu = load_clip_to_cpu
    net = slinet.SliNet(ref_args(cfg))
    for t in range(len(net.prompts)):
        fac = synth.prompt_factors(9, 16, cfg.vision_width, cfg.transformer_width, task=t)
        for k, v in fac.items():
            getattr(net.prompts[t], k).data = torch.from_numpy(v.copy())
    return net


def patch_depth(depth):
    """Re-state ResidualAttentionBlock.forward with the intended guard (model.py:187-196)."""
    from models.clip import model as M

    orig = M.ResidualAttentionBlock.forward

    def forward(self, inp):
        x, prompts = inp[0], inp[1]
        if 0 < self.layer_id < depth and prompts is not None:
            P = prompts.shape[-2]
            tmp = prompts[:, self.layer_id, :, :].permute(1, 0, 2)
            x = torch.cat([x[:1], x[1:P + 1] + tmp, x[P + 1:]], dim=0)
        x = x + self.attention(self.ln_1(x))
        x = x + self.mlp(self.ln_2(x))
        return [x, prompts]

    M.ResidualAttentionBlock.forward = forward
    return lambda: setattr(M.ResidualAttentionBlock, "forward", orig)


CAPTIONS = [
    "a man riding a wave on top of a surfboard",
    "two dogs play with a red frisbee in the park",
    "a plate of food with broccoli and rice",
    "an old clock tower stands over the city street",
    "a woman holding an umbrella while it rains",
    "several giraffes eating leaves from tall trees",
    "a kitchen with stainless steel appliances and a wooden table",
    "a baseball player swings his bat at the ball",
    "a cat sleeping on a laptop keyboard",
    "people walking through a busy train station at night",
    "a red double decker bus driving down a street",
    "a small child flying a colorful kite on the beach",
]


def captions_for(batch):
    return [CAPTIONS[i % len(CAPTIONS)] + ("" if i < len(CAPTIONS) else f" number {i}") for i in range(batch)]


def train_step(net, cfg, batch, numtask):
    """SliNet.forward + cal_loss + backward exactly as SPrompts.train_function does (sprompt.py:297-311)."""
    net.numtask = numtask
    net.train()
    names = []
    for name, p in net.named_parameters():
        p.requires_grad_(False)
        if "prompts." + str(numtask - 1) + "." in name:  # sprompt.py:235
            p.requires_grad_(True)
            names.append(name)
            p.grad = None
    img = torch.from_numpy(synth.images(batch, cfg.image_resolution))
    caps = captions_for(batch)
    img_f, txt_f, vp, tp = net(img, caps)
    out = net.cal_loss(img_f, txt_f, vp, tp)
    loss = sum(v for v in out["loss"].values())
    loss.backward()
    with torch.no_grad():
        logits = net.logit_scale.exp() * img_f @ txt_f.t()
        ids = torch.cat([__import__("models.clip.clip", fromlist=["tokenize"]).tokenize(
            " ".join(["X"] * 16) + " " + c + ".") for c in caps])
    res = {
        "token_ids": ids.numpy(),
        "img_f": img_f.detach().float().numpy(), "txt_f": txt_f.detach().float().numpy(),      # (.float(): the fp16 family stores what fp16 held, widened)
        "logits": logits.float().numpy(),
        "vis_prompt": vp[0].detach().float().numpy(), "txt_prompt": tp[0].detach().float().numpy(),
        "trainable": np.array(names),
    }
    for k, v in out["loss"].items():
        res[k] = np.float32(v.item())
    for name, p in net.named_parameters():
        if p.requires_grad:
            res["grad." + name.split(".")[-1]] = p.grad.detach().float().numpy().copy()
    # per-row top-5 with margins (F8): index parity is only asserted where the margin is large
    for tag, S in (("i2t", logits), ("t2i", logits.t())):
        srt, idx = torch.sort(S, dim=1, descending=True, stable=True)
        k = min(5, S.shape[1] - 1)
        res[f"top5_{tag}"] = idx[:, :k].numpy()
        res[f"top5_margin_{tag}"] = (srt[:, :k] - srt[:, 1:k + 1]).numpy()
    return res


def train_step_ids(net, cfg, batch, numtask, ids):
    """The same step as `train_step`, on TOKEN IDS given by the caller (round 6: bench.py's own batch, `synth.token_ids(256)`, whose random ids are not the
    BPE of any string).  `clip.tokenize` as PromptLearner.forward reaches it (prompt_learner.py:131: one call per prompt) is replaced by a function that
    hands out the next prepared row; everything else — embedding look-up, ctx insertion, both towers, cal_loss, backward — is the imported reference.
    Tokenisation itself is pinned separately (tokenizer.npz / tokenizer_wide.npz)."""
    import models.clip.prompt_learner as PL
    rows = [torch.from_numpy(ids[i:i + 1].copy()) for i in range(batch)]
    it = iter(rows)
    real = PL.clip.tokenize
    PL.clip.tokenize = lambda *_a, **_k: next(it)
    try:
        net.numtask = numtask
        net.train()
        names = []
        for name, p in net.named_parameters():
            p.requires_grad_(False)
            if "prompts." + str(numtask - 1) + "." in name:  # sprompt.py:235
                p.requires_grad_(True)
                names.append(name)
                p.grad = None
        img = torch.from_numpy(synth.images(batch, cfg.image_resolution))
        img_f, txt_f, vp, tp = net(img, [f"caption {i}" for i in range(batch)])
        out = net.cal_loss(img_f, txt_f, vp, tp)
        loss = sum(v for v in out["loss"].values())
        loss.backward()
    finally:
        PL.clip.tokenize = real
    with torch.no_grad():
        logits = net.logit_scale.exp() * img_f @ txt_f.t()
    res = {"token_ids_crc32": np.uint32(__import__("zlib").crc32(np.ascontiguousarray(ids).tobytes())),      # the ids are regenerated from the seed by the tests
           "img_f": img_f.detach().numpy(), "txt_f": txt_f.detach().numpy(), "logits": logits.numpy(), "trainable": np.array(names)}
    for k, v in out["loss"].items():
        res[k] = np.float32(v.item())
    for name, p in net.named_parameters():
        if p.requires_grad:
            res["grad." + name.split(".")[-1]] = p.grad.detach().numpy().copy()
    for tag, S in (("i2t", logits), ("t2i", logits.t())):
        srt, idx = torch.sort(S, dim=1, descending=True, stable=True)
        res[f"top5_{tag}"] = idx[:, :5].numpy().astype(np.int32)
        res[f"top5_margin_{tag}"] = (srt[:, :5] - srt[:, 1:6]).numpy()
    return res


def task_loss_case(cfg=None):
    """`SliNet.cal_task_loss` (slinet.py:167-183) -> `nt_bxent_loss` (loss/loss.py:6-33) at the sizes the reference runs it: the stacked, flattened prompts of
    tasks 0..t are [t+1, 9*16*768 = 110 592] and [t+1, 73 728], temperature 0.001, for numtask in {2, 7, 12}.  The imported method is called unbound on a
    stand-in that holds only what it reads (`self.prompts`: the imported DecomposedPrompt modules) — no backbone is involved.  Two families of factors:
    'random' = the reference's own initialiser scale (synth.prompt_factors(task=t): pairwise cosines of a few 1e-3, i.e. cos / 0.001 of order one — the
    regime where the double sigmoid decides the loss), 'drift' = every task a small step away from a common ancestor (what a continual session that
    initialises from similar solutions looks like: cosines near one, the inner sigmoid saturated and the gradient exactly zero), 'mixed' = tasks 0-3 drifting,
    the later ones independent.  Stored: the loss and the gradient w.r.t. the CURRENT
    task's five factors (sprompt.py:235: the only trainable ones)."""
    import models.slinet as slinet
    from models.prompts.prompts import DecomposedPrompt
    cfg = cfg or synth.VIT_B16
    res = {"widths": np.array([cfg.vision_width, cfg.transformer_width]), "numtasks": np.array([2, 7, 12])}
    for fam in ("random", "drift", "mixed"):
        mods = []
        for t in range(12):
            m = DecomposedPrompt(9, 16, cfg.vision_width, cfg.transformer_width)
            fac = synth.task_family_factors(fam, t, cfg.vision_width, cfg.transformer_width)
            for k, v in fac.items():
                getattr(m, k).data = torch.from_numpy(v.copy())
            mods.append(m)
        ns = types.SimpleNamespace(prompts=mods)
        for numtask in (2, 7, 12):
            for m in mods:
                for p_ in m.parameters():
                    p_.requires_grad_(False)
                    p_.grad = None
            for p_ in mods[numtask - 1].parameters():
                p_.requires_grad_(True)
            dummy = torch.zeros(1)
            loss = slinet.SliNet.cal_task_loss(ns, numtask - 1, dummy, dummy)
            loss.backward()
            res[f"{fam}.{numtask}.loss"] = np.float32(loss.item())
            for k in synth.PROMPT_NAMES:
                res[f"{fam}.{numtask}.grad.{k}"] = getattr(mods[numtask - 1], k).grad.numpy().copy()
            with torch.no_grad():      # the cosines themselves, for the record of which regime a case is in
                for tag, j in (("v", 0), ("t", 1)):
                    X = torch.stack([mods[i]()[j].view(-1) for i in range(numtask)])
                    Xn = X / X.norm(dim=-1, keepdim=True)
                    res[f"{fam}.{numtask}.cos_{tag}"] = (Xn @ Xn.t()).numpy()
    return res


def tokenizer_wide_texts(n=1200, seed=606):
    """>= 1000 strings for the tokenizer fixture (round 6): COCO-like captions from templates, and the things captions in the wild carry — HTML entities
    (html.unescape runs twice in basic_clean, simple_tokenizer.py:50-53), runs of whitespace / tabs / newlines (whitespace_clean), mixed case, digits and
    decimals, contractions (the pattern's 's|'t|'re|'ve|'m|'ll|'d alternatives), hyphens and punctuation runs, Latin-1 / Greek / Cyrillic / Hebrew / Arabic /
    CJK / Hangul / Thai text, emoji incl. ZWJ sequences and skin tones, the special-token strings, very long words, and > 77-token overflows."""
    import random
    rng = random.Random(seed)
    subj = ["a man", "a woman", "two dogs", "a group of people", "three children", "an old bus", "the black cat", "a young girl", "several giraffes",
            "a baseball player", "a red double decker bus", "the chef", "a flock of birds", "someone", "a surfer", "two zebras", "an elderly couple"]
    verb = ["riding", "holding", "standing next to", "looking at", "sitting on", "eating", "walking past", "flying", "parked near", "jumping over",
            "playing with", "waiting for", "leaning against", "cutting", "pointing at"]
    obj = ["a wave", "an umbrella", "a wooden table", "a plate of broccoli and rice", "the train station", "a laptop keyboard", "a colorful kite",
           "a fire hydrant", "a stop sign", "a frisbee", "some tall trees", "a pizza", "the city street", "a tennis racket", "a pile of luggage"]
    tail = ["", " in the park", " at night", " on a sunny day", " near the beach", " in black and white", " while it rains", " with a blue sky behind",
            " during the 2014 world cup", " at 5:30 pm", " for $3.50", " (blurry)", " -- taken in 1998", " #nofilter", ", isn't it?", "!!!", "..."]
    contr = ["don't", "it's", "we're", "I've", "I'm", "they'll", "he'd", "can't", "won't", "'tis", "o'clock", "dog's", "dogs'", "y'all'd've", "'s", "'T"]
    ent = ["&amp;", "&lt;b&gt;", "&quot;quoted&quot;", "&#39;", "&amp;amp;", "&nbsp;", "&eacute;", "&copy; 2014", "&lt;|endoftext|&gt;", "&#x1F600;"]
    ws = ["  ", "   ", "\t", "\n", " \n ", "\r\n", "\u00a0", "\u2003"]
    scripts = ["caf\u00e9 na\u00efve r\u00e9sum\u00e9 \u00fcber stra\u00dfe", "\u03b1\u03b2\u03b3 \u0394\u03b5\u03bb\u03c4\u03b1", "\u0416\u0434\u0451\u043c \u043f\u043e\u0435\u0437\u0434",
               "\u05e9\u05dc\u05d5\u05dd \u05e2\u05d5\u05dc\u05dd", "\u0645\u0631\u062d\u0628\u0627 \u0628\u0627\u0644\u0639\u0627\u0644\u0645", "\u4f60\u597d\uff0c\u4e16\u754c\uff01", "\u3053\u3093\u306b\u3061\u306f \u4e16\u754c",
               "\ud55c\uad6d\uc5b4 \ud14c\uc2a4\ud2b8", "\u0e2a\u0e27\u0e31\u0e2a\u0e14\u0e35", "\u0967\u0968\u0969 \u0664\u0665\u0666 \uff11\uff12\uff13 \u2163 \u00bd \u00b2", "\u00bfqu\u00e9? \u00a1s\u00ed! \u20ac5 \u00a33 \u00a5100 \u00b0C \u00b15%"]
    emoji = ["\U0001F600", "\U0001F389\U0001F389", "\U0001F44D\U0001F3FD", "\u2708\ufe0f", "\U0001F468\u200d\U0001F469\u200d\U0001F467", "\u2764\ufe0f\u200d\U0001F525", "\U0001F1EF\U0001F1F5", "\u263a"]
    special = ["<|startoftext|>", "<|endoftext|>", "hello<|endoftext|>world", "<|startoftext|> a cat <|endoftext|>", "<|STARTOFTEXT|>", "<|endoftext|"]
    out = []

    def cap():
        return f"{rng.choice(subj)} {rng.choice(verb)} {rng.choice(obj)}{rng.choice(tail)}"

    def mixcase(t):
        return "".join(c.upper() if rng.random() < 0.4 else c for c in t)
    for i in range(n):
        k = i % 12
        if k == 0:
            t = cap()
        elif k == 1:
            t = mixcase(cap())
        elif k == 2:
            t = cap().replace(" ", rng.choice(ws), rng.randint(1, 4)) + rng.choice(ws)
        elif k == 3:
            t = f"{cap()} {rng.choice(ent)} {rng.choice(ent)}{rng.choice(obj)}"
        elif k == 4:
            t = f"{rng.choice(contr)} {cap()} {rng.choice(contr)} {rng.choice(contr)}"
        elif k == 5:
            t = f"{rng.randint(0, 99999)} {rng.random() * 1000:.3f} {rng.randint(0, 9)}x{rng.randint(0, 9)} {cap()} no.{i} 1,000,000 3rd 0x{i:04X}"
        elif k == 6:
            t = f"{rng.choice(scripts)} {cap()} {rng.choice(scripts)}"
        elif k == 7:
            t = f"{cap()} {rng.choice(emoji)}{rng.choice(emoji)} {rng.choice(emoji)}"
        elif k == 8:
            t = f"{rng.choice(special)} {cap()}" if rng.random() < 0.5 else rng.choice(special)
        elif k == 9:
            t = "".join(rng.choice("abcdefghijklmnopqrstuvwxyz") for _ in range(rng.randint(20, 60))) + " " + "-".join(rng.choice(obj).split()) + "_" * rng.randint(1, 5) + "/\\|~^`"
        elif k == 10:
            t = " ".join(cap() for _ in range(rng.randint(4, 9)))          # > 77 tokens: overflow
        else:
            t = rng.choice(["", " ", ".", "a", "A.", "?!", "''", "' '", "\t\n", "x" * rng.randint(1, 90), "X X X X X X X X X X X X X X X X " + cap() + "."])
        out.append(t)
    return out


def tokenizer_wide_case():
    """`SimpleTokenizer.encode` (simple_tokenizer.py:121-127) and `clip.tokenize` (clip.py:185-221) of the imported reference on tokenizer_wide_texts():
    the ragged ids of every string (encode), the padded [n, 77] rows of those that fit, and which ones raise the too-long RuntimeError (clip.py:213-218).
    (`ftfy.fix_text` is the identity in this harness — ftfy is not installed — and in this repo's tokenizer alike.)"""
    from models.clip.clip import tokenize
    from models.clip.clip import _tokenizer as tk
    texts = tokenizer_wide_texts()
    enc = [tk.encode(t) for t in texts]
    offs = np.cumsum([0] + [len(e) for e in enc]).astype(np.int64)
    flat = np.array([i for e in enc for i in e], dtype=np.int32)
    too_long = np.zeros(len(texts), dtype=bool)
    rows = np.zeros((len(texts), 77), dtype=np.int32)
    for i, t in enumerate(texts):
        try:
            rows[i] = tokenize(t)[0].numpy()
        except RuntimeError:
            too_long[i] = True
            rows[i] = tokenize(t, truncate=True)[0].numpy()
    assert too_long.sum() >= 50 and (~too_long).sum() >= 1000
    return {"texts": np.array(texts), "encode_flat": flat, "encode_offsets": offs, "too_long": too_long, "rows": rows}


def eval_case(net, cfg, batch):
    """Eval interfaces (slinet.py:85-107, 185-220) + task-id selection and itm_eval (sprompt.py:336-368, 550-646)."""
    from methods.sprompt import SPrompts

    net.numtask = 3
    net.eval()
    img = torch.from_numpy(synth.images(batch, cfg.image_resolution, seed=synth.IMAGE_SEED + 7))
    caps = captions_for(batch)
    rng = np.random.Generator(np.random.Philox(key=[77, 1]))
    sel_v = torch.from_numpy(rng.integers(0, 3, size=batch))
    sel_t = torch.from_numpy(rng.integers(0, 3, size=batch))
    with torch.no_grad():
        ev = net.extract_vector(img)
        et = net.extract_textual_vector(caps)
        vi = net.visual_interface(img, sel_v)
        ti = net.textual_interface(caps, sel_t)
    from models.clip.clip import tokenize
    ids = torch.cat([tokenize(" ".join(["X"] * 16) + " " + c + ".") for c in caps])
    res = {"token_ids": ids.numpy(), "captions": np.array(caps),
           "sel_v": sel_v.numpy(), "sel_t": sel_t.numpy(), "extract_vector": ev.numpy(),
           "extract_textual_vector": et.numpy(), "visual_interface": vi.numpy(),
           "textual_interface": ti.numpy()}

    # task-id selection by L1 distance to per-task centres (sprompt.py:336-351)
    sp = object.__new__(SPrompts)
    sp._network = net
    keys = [torch.from_numpy(synth.normal(5, f"keys{t}", (5, cfg.embed_dim), 0.05)) for t in range(3)]
    sp.all_keys = keys
    sp.textual_all_keys = keys
    res["task_keys"] = torch.stack(keys).numpy()
    res["visual_task_id"] = sp.get_visual_task_id(img).numpy()
    res["textual_task_id"] = sp.get_textual_task_id(caps).numpy()

    # itm_eval on a fixed score matrix: 2 captions per image, 3 tasks
    n_img, n_txt = 24, 48
    scores = synth.normal(9, "scores", (n_img, n_txt))
    img2txt = {i: [2 * i, 2 * i + 1] for i in range(n_img)}
    txt2img = {t: t // 2 for t in range(n_txt)}
    cat_i = [i % 3 for i in range(n_img)]
    cat_t = [(t // 2) % 3 for t in range(n_txt)]
    sp.cur_id = 2
    fr = sp.itm_eval(scores, scores.T.copy(), txt2img, img2txt, cat_i, torch.tensor(cat_t))
    res["itm_scores"] = scores
    res["itm_cat_i"] = np.array(cat_i)
    res["itm_cat_t"] = np.array(cat_t)
    res["itm_i2t"] = np.array([fr["mscoco"]["i2t"][t] for t in range(3)])
    res["itm_t2i"] = np.array([fr["mscoco"]["t2i"][t] for t in range(3)])
    return res


def sgd_steps(net, cfg, batch, steps):
    """`steps` epochs of ONE batch each through the reference's optimiser set-up (sprompt.py:230-255: trainable filter, SGD(momentum .9,
    lr, weight_decay) over network.parameters(), CosineAnnealingLR(T_max=epochs) stepped per epoch, :324) and hot loop (:297-311)."""
    from torch import optim
    net.numtask = 1
    net.train()
    for name, p in net.named_parameters():
        p.requires_grad_(False)
        if "prompts." + str(net.numtask - 1) + "." in name:
            p.requires_grad_(True)
    args = ref_args(cfg)
    opt = optim.SGD(net.parameters(), momentum=0.9, lr=args["lrate"], weight_decay=args["weight_decay"])
    sched = optim.lr_scheduler.CosineAnnealingLR(optimizer=opt, T_max=steps)
    img = torch.from_numpy(synth.images(batch, cfg.image_resolution))
    caps = captions_for(batch)
    res = {"lrate": np.float32(args["lrate"]), "weight_decay": np.float32(args["weight_decay"]), "steps": np.int32(steps)}
    for st in range(steps):
        img_f, txt_f, vp, tp = net(img, caps)
        out = net.cal_loss(img_f, txt_f, vp, tp)
        loss = sum(v for v in out["loss"].values())
        opt.zero_grad()
        loss.backward()
        opt.step()
        sched.step()
        res[f"loss.{st}"] = np.float32(loss.item())
        res[f"lr_after.{st}"] = np.float32(opt.param_groups[0]["lr"])
        for name, p in net.named_parameters():
            if p.requires_grad:
                res[f"param.{st}.{name.split('.')[-1]}"] = p.detach().numpy().copy()
    # restore the factors for whoever uses the net next
    fac = synth.prompt_factors(9, 16, cfg.vision_width, cfg.transformer_width, task=0)
    for k, v in fac.items():
        getattr(net.prompts[0], k).data = torch.from_numpy(v.copy())
    return res


def eval_shard(net, cfg, n_img, caps_per_img, n_tasks):
    """The reference's whole retrieval evaluation (sprompt.py:433-548 `_evaluate_retrieval` -> :550-646 `itm_eval`) on a fixed synthetic
    shard: task-id selection by L1 distance to per-task keys, per-sample prompted features, the N_img x N_txt score matrix, R@K — plus,
    for index parity (F8), the per-row rank of the best ground truth under np.argsort(score)[::-1] and the margin that makes it safe."""
    from methods.sprompt import SPrompts
    from models.clip.clip import tokenize

    net.numtask = n_tasks
    net.eval()
    n_txt = n_img * caps_per_img
    img = torch.from_numpy(synth.images(n_img, cfg.image_resolution, seed=synth.IMAGE_SEED + 11))
    words = ["red", "blue", "green", "small", "large", "old", "young", "wooden", "bright", "dark", "quiet", "busy"]
    caps = [CAPTIONS[(t // caps_per_img) % len(CAPTIONS)] + f" {words[t % len(words)]} view {t}" for t in range(n_txt)]
    cat_i = [i % n_tasks for i in range(n_img)]
    cat_t = [(t // caps_per_img) % n_tasks for t in range(n_txt)]

    class DS:
        pass

    ds = DS()
    ds.text, ds.image, ds.text_cat = caps, list(range(n_img)), cat_t
    ds.img2txt = {i: [caps_per_img * i + j for j in range(caps_per_img)] for i in range(n_img)}
    ds.txt2img = {t: t // caps_per_img for t in range(n_txt)}

    class Loader:
        batch_size = 16
        dataset = ds

        def __iter__(self):
            for i in range(0, n_img, self.batch_size):
                j = min(n_img, i + self.batch_size)
                yield img[i:j], torch.arange(i, j), torch.tensor(cat_i[i:j])

    sp = object.__new__(SPrompts)
    sp._network, sp.args, sp._device, sp._multiple_gpus, sp.cur_id = net, {"prompt_type": "lpi"}, torch.device("cpu"), [], n_tasks - 1
    with torch.no_grad():
        ev = net.extract_vector(img)            # un-prompted features (the task-id pass, sprompt.py:336-351)
        et = net.extract_textual_vector(caps)
    # keys: five un-prompted features of each task's own samples (so that those samples select their task with a wide margin)
    vkeys = [ev[[i for i in range(n_img) if cat_i[i] == t][:5]].clone() for t in range(n_tasks)]
    tkeys = [et[[i for i in range(n_txt) if cat_t[i] == t][:5]].clone() for t in range(n_tasks)]
    sp.all_keys, sp.textual_all_keys = vkeys, tkeys
    s_i2t, s_t2i, final_res = sp._evaluate_retrieval(Loader())
    with torch.no_grad():
        sel_v, sel_t = sp.get_visual_task_id(img), sp.get_textual_task_id(caps)
        vfeat = net.visual_interface(img, sel_v)
        tfeat = net.textual_interface(caps, sel_t)

    def l1(f, keys):
        return torch.stack([torch.stack([(f - c).abs().sum(1) for c in k]).min(0)[0] for k in keys], 1).numpy()

    def ranks(S, gts):
        r, m = np.zeros(len(S), np.int32), np.zeros(len(S), np.float32)
        for i, row in enumerate(S):
            inds = np.argsort(row)[::-1]
            r[i] = min(int(np.where(inds == g)[0][0]) for g in gts[i])
            others = np.delete(row, gts[i])
            m[i] = min(float(np.abs(others - row[g]).min()) for g in gts[i])      # the rank is safe if no other score is this close
        return r, m

    r_i, m_i = ranks(s_i2t, [ds.img2txt[i] for i in range(n_img)])
    r_t, m_t = ranks(s_t2i, [[ds.txt2img[t]] for t in range(n_txt)])
    ids = torch.cat([tokenize(" ".join(["X"] * 16) + " " + c + ".") for c in caps])
    return {"captions": np.array(caps), "token_ids": ids.numpy(), "cat_i": np.array(cat_i), "cat_t": np.array(cat_t),
            "caps_per_img": np.int32(caps_per_img), "n_tasks": np.int32(n_tasks),
            "vkeys": torch.stack(vkeys).numpy(), "tkeys": torch.stack(tkeys).numpy(),
            "extract_vector": ev.numpy(), "extract_textual_vector": et.numpy(),
            "visual_task_dist": l1(ev, vkeys), "textual_task_dist": l1(et, tkeys),
            "visual_task_id": sel_v.numpy(), "textual_task_id": sel_t.numpy(),
            "image_feats": vfeat.numpy(), "text_feats": tfeat.numpy(), "score_i2t": s_i2t.astype(np.float32),
            "rank_i2t": r_i, "rank_margin_i2t": m_i, "rank_t2i": r_t, "rank_margin_t2i": m_t,
            "itm_i2t": np.array([final_res["mscoco"]["i2t"][t] for t in range(n_tasks)]),
            "itm_t2i": np.array([final_res["mscoco"]["t2i"][t] for t in range(n_tasks)])}


def kmeans_case(n=600, dim=512):
    """The reference's task-key clustering (methods/sprompt.py:370-397 `clustering`: un-prompted features of the task's training set, L2-normalised,
    KMeans(n_clusters=5, random_state=0) per modality; scikit-learn as installed here) run through the IMPORTED method on synthetic features
    (lpi_amd.synth.clustering_features): the network is a stand-in that returns rows of the feature matrices, everything else is the reference's code."""
    from methods.sprompt import SPrompts
    import sklearn
    fv, ft = (torch.from_numpy(x) for x in synth.clustering_features(n, dim))

    class Net:
        def extract_vector(self, idx):
            return fv[idx]

        def extract_textual_vector(self, idx):
            return ft[torch.as_tensor(idx)]

    class Loader:
        def __iter__(self):
            for i in range(0, n, 128):
                idx = torch.arange(i, min(n, i + 128))
                yield idx, idx.tolist(), None, None

    sp = object.__new__(SPrompts)
    sp._network, sp._device, sp.all_keys, sp.textual_all_keys = Net(), torch.device("cpu"), [], []
    sp.clustering(Loader())
    return {"shape": np.array([n, dim]), "centers_visual": sp.all_keys[0].numpy(), "centers_textual": sp.textual_all_keys[0].numpy(),
            "sklearn_version": np.array(sklearn.__version__)}


def save(name, res, meta):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **res)
    meta[name] = {k: (list(v.shape) if hasattr(v, "shape") else None) for k, v in res.items()}
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.0f} KiB)")


def interact_case(bs=2, P=16, Dv=96, Dt=768, layer_num=9, r=4, layer_id=3):
    """The grounding branch's InteractModule (modeling_bert.py:558-651) imported as it is — two names its module imports from `transformers` are gone
    from the installed version and are stubbed (ordinary ImportError shims, like the retrieval ones above) — on [bs, P, D] prompt rows: outputs and
    the gradients of sum(v_out * Wv) + sum(t_out * Wt) w.r.t. every parameter and both inputs."""
    import types
    import transformers.pytorch_utils as pu
    for n in ("apply_chunking_to_forward", "find_pruneable_heads_and_indices", "prune_linear_layer"):
        if not hasattr(pu, n):
            setattr(pu, n, lambda *a_, **k_: None)
    if "transformers.onnx" not in sys.modules:
        onnx = types.ModuleType("transformers.onnx")
        onnx.OnnxConfig = object
        sys.modules["transformers.onnx"] = onnx
    sys.path.insert(0, os.path.join(os.path.dirname(REF), "grounding"))
    from maskrcnn_benchmark.modeling.bert.modeling_bert import InteractModule
    torch.manual_seed(20240607)
    m = InteractModule(layer_num=layer_num, visual_dim=Dv, textual_dim=Dt, r=r)
    with torch.no_grad():      # LayerNorm affine away from the identity, so that its gradients are exercised
        for ln in (m.visual_norm, m.textual_norm):
            ln.weight.add_(0.1 * torch.randn_like(ln.weight))
            ln.bias.add_(0.1 * torch.randn_like(ln.bias))
    inp = synth.interact_inputs(bs, P, Dv, Dt)          # regenerated from the seed by the tests: not stored
    v = torch.from_numpy(inp["visual_in"]).requires_grad_(True)
    t = torch.from_numpy(inp["textual_in"]).requires_grad_(True)
    wv, wt = torch.from_numpy(inp["wv"]), torch.from_numpy(inp["wt"])
    vo, to = m(v, t, layer_id)
    ((vo * wv).sum() + (to * wt).sum()).backward()
    res = {"layer_id": np.int64(layer_id), "shape": np.array([bs, P, Dv, Dt, layer_num, r]),
           "visual_out": vo.detach().numpy(), "textual_out": to.detach().numpy(), "grad.visual_in": v.grad.numpy(), "grad.textual_in": t.grad.numpy()}
    for n, p_ in m.named_parameters():
        res["param." + n] = p_.detach().numpy().copy()
        res["grad." + n] = p_.grad.numpy().copy()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    meta_path = os.path.join(HERE, "MANIFEST.json")
    meta = json.load(open(meta_path)) if os.path.exists(meta_path) else {}
    if a.only in (None, "interact"):      # before the retrieval shims: `transformers` probes the real torchvision's module spec
        save("interact", interact_case(), meta)
    install_shims()
    meta["_generator"] = {"torch": torch.__version__, "numpy": np.__version__, "threads": torch.get_num_threads(),
                          "reference": "Kelvin-ywc/LPI @ 2024-12-23, retrieval/", "dtype": "float32 (CPU)"}

    if a.only in (None, "kmeans"):
        save("kmeans", kmeans_case(), meta)

    if a.only in (None, "tokenizer"):
        from models.clip.clip import tokenize
        texts = ["A photo of a cat.", "two  dogs,   running!", "it's the man's 3rd try -- isn't it?", "caf\u00e9 na\u00efve r\u00e9sum\u00e9",
                 "12345 67 890", "hello<|endoftext|>world", "UPPER lower MiXeD", "a/b c-d e_f (g) [h] {i}", "\u4f60\u597d \u4e16\u754c",
                 "emoji \U0001F600 test", "x" * 40, "the quick brown fox jumps over the lazy dog " * 3, "&amp; &lt;tag&gt; &quot;q&quot;",
                 " ".join(["X"] * 16) + " a man riding a wave on top of a surfboard."]
        save("tokenizer", {"texts": np.array(texts), "ids": tokenize(texts).numpy()}, meta)

    if a.only in (None, "tokenizer", "tokenizer_wide"):
        save("tokenizer_wide", tokenizer_wide_case(), meta)

    if a.only in (None, "tiny"):
        cfg = synth.TINY
        net = build_slinet(cfg)
        save("tiny_d1", train_step(net, cfg, 4, 1), meta)
        save("tiny_task2", train_step(net, cfg, 4, 2), meta)     # exercises task_loss (numtask != 1)
        undo = patch_depth(2)
        save("tiny_d2_patched", train_step(net, cfg, 4, 1), meta)   # tiny has 2 layers: depth 2 is the deep case
        undo()
        save("tiny_eval", eval_case(net, cfg, 6), meta)

    if a.only in (None, "tiny", "tiny_sgd3"):
        cfg = synth.TINY
        net = build_slinet(cfg)
        save("tiny_sgd3", sgd_steps(net, cfg, 4, 3), meta)       # post-step parameters of 3 SGD + cosine steps (a10)
        save("tiny_eval_shard", eval_shard(net, cfg, 12, 2, 3), meta)

    if a.only in (None, "vitb16"):
        cfg = synth.VIT_B16
        net = build_slinet(cfg)
        r = train_step(net, cfg, 8, 1)          # BASELINE.json configs[0]: bs=8, r=4
        for k in ("vis_prompt", "txt_prompt"):  # keep the fixture small: layer 0 only (the live one, F1)
            r[k] = r[k][:1]
        save("vitb16_d1", r, meta)
        undo = patch_depth(3)
        r = train_step(net, cfg, 8, 1)
        for k in ("vis_prompt", "txt_prompt"):
            r[k] = r[k][:3]
        save("vitb16_d3_patched", r, meta)
        undo()


    if a.only in (None, "task_loss"):      # round 6: a9 at its operating size (12 tasks, real widths, temperature 0.001)
        save("task_loss_wide", task_loss_case(), meta)
    if a.only in (None, "vitb16", "vitb16_task12"):      # the whole step of the LAST task of a 12-task session (task term included), ViT-B/16, 8 pairs
        cfg = synth.VIT_B16
        net = build_slinet(cfg)
        r = train_step(net, cfg, 8, 12)
        for k in ("vis_prompt", "txt_prompt"):
            r[k] = r[k][:1]
        save("vitb16_task12", r, meta)

    if a.only in (None, "fp16"):
        # the reference in its own fp16 (convert_weights), torch CPU: tiny and ViT-B/16 (8 pairs), depth 1 = the shipped code
        for name, cfg, batch in (("tiny_fp16", synth.TINY, 4), ("vitb16_fp16", synth.VIT_B16, 8)):
            net = build_slinet(cfg, fp16=True)
            assert net.dtype == torch.float16
            r = train_step(net, cfg, batch, 1)
            for k in ("vis_prompt", "txt_prompt"):
                r[k] = r[k][:1]
            r["reference_dtype"] = np.array("float16 (convert_weights), torch CPU")
            save(name, r, meta)


    # round 6: the BENCHMARKED configuration at its own size (BASELINE.json configs[2]: ViT-B/16, 256 pairs; sprompt.py:297-311) on bench.py's own
    # synthetic batch (synth.images(256), synth.token_ids(256)): ~35 GB of autograd state, a few minutes of CPU each; not part of the default run.
    if a.only in ("bs256", "bs256_d3"):
        cfg = synth.VIT_B16
        net = build_slinet(cfg)
        undo = patch_depth(3)
        save("vitb16_bs256_d3_patched", train_step_ids(net, cfg, 256, 1, synth.token_ids(256)), meta)
        undo()
    if a.only in ("bs256", "bs256_d1"):
        cfg = synth.VIT_B16
        net = build_slinet(cfg)
        save("vitb16_bs256_d1", train_step_ids(net, cfg, 256, 1, synth.token_ids(256)), meta)

    if a.only in (None, "vitb16", "vitb16_eval"):
        cfg = synth.VIT_B16
        net = build_slinet(cfg)
        save("vitb16_eval", eval_shard(net, cfg, 32, 2, 3), meta)   # north_star: R@1 indices on a fixed synthetic shard at full size

    if a.only in ("vitb16_eval12",):      # round 6: 256 images x 1 280 captions x 12 tasks (minutes of CPU: not part of the default run)
        cfg = synth.VIT_B16
        net = build_slinet(cfg)
        r = eval_shard(net, cfg, 256, 5, 12)
        for k in ("extract_textual_vector", "text_feats"):      # 2 x 2.6 MB that the score matrix (image_feats x text_feats) and the L1 distances already pin
            r.pop(k)
        r["token_ids"] = r["token_ids"].astype(np.int32)
        save("vitb16_eval12", r, meta)

    with open(meta_path, "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()

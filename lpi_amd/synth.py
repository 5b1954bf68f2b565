"""Deterministic synthetic CLIP weights, prompt factors and inputs.

Real CLIP ViT-B/16 weights cannot be fetched (no network; the reference downloads them in
``retrieval/models/clip/prompt_learner.py:10-13`` / ``clip.py:39-68``).  Everything here is drawn from
numpy's Philox bit generator keyed by ``(seed, crc32(tensor name))`` so that the golden generator
(which loads the tensors into the imported reference in the build container) and the GPU box
(which has no reference) regenerate bit-identical tensors without shipping any file.

Scales follow the reference's own initialiser ``CLIP.initialize_parameters``
(``retrieval/models/clip/model.py:318-345``) so activations are well conditioned; biases and LayerNorm
affine terms get small non-trivial values so every term of every kernel is exercised.

The state-dict key names are the reference's (``model.py:418-441`` infers the architecture from them),
so a real CLIP state dict can be passed to the engine in place of the synthetic one.
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass, asdict

import numpy as np

WEIGHT_SEED = 20240101
PROMPT_SEED = 3000
IMAGE_SEED = 1000               # + rank
TOKEN_SEED = 2000               # + rank

SOT, EOT, X_TOKEN, DOT_TOKEN = 49406, 49407, 343, 269   # [probed] ids of "<|startoftext|>", "<|endoftext|>", "x</w>", ".</w>"


@dataclass(frozen=True)
class ClipConfig:
    """Architecture numbers exactly as ``CLIP.__init__`` takes them (``model.py:262-277``)."""
    name: str
    embed_dim: int
    image_resolution: int
    vision_layers: int
    vision_width: int
    vision_patch_size: int
    context_length: int
    vocab_size: int
    transformer_width: int
    transformer_heads: int
    transformer_layers: int

    @property
    def vision_heads(self) -> int:
        return self.vision_width // 64

    @property
    def n_patches(self) -> int:
        return (self.image_resolution // self.vision_patch_size) ** 2

    def as_clip_args(self):
        d = asdict(self)
        d.pop("name")
        return tuple(d.values())


# head_dim is 64 for every CLIP ViT (vision_heads = width // 64, model.py:292)
VIT_B16 = ClipConfig("ViT-B/16", 512, 224, 12, 768, 16, 77, 49408, 512, 8, 12)
VIT_L14 = ClipConfig("ViT-L/14", 768, 224, 24, 1024, 14, 77, 49408, 768, 12, 12)
VIT_L14_336 = ClipConfig("ViT-L/14@336px", 768, 336, 24, 1024, 14, 77, 49408, 768, 12, 12)      # 577 vision tokens: the long-sequence attention kernels
VIT_B32 = ClipConfig("ViT-B/32", 512, 224, 12, 768, 32, 77, 49408, 512, 8, 12)      # clip.available_models(): 50 vision tokens, 3 072-column patches
# small config used by fast tests: 2 layers, 2 heads, 4 patches; same code paths
TINY = ClipConfig("tiny", 128, 32, 2, 128, 16, 77, 49408, 128, 2, 2)

# patch 14 (K = 588, zero padded to the GEMM K tile) like ViT-L/14, at toy size
TINY14 = ClipConfig("tiny14", 128, 28, 2, 256, 14, 77, 49408, 128, 2, 2)   # widths must be multiples of 128 (GEMM tile)

# 401 vision tokens (+ prompts) at toy width: the long-sequence attention kernels (csrc/attn_long.hip; ViT-L/14@336px has 577)
TINY_LONG = ClipConfig("tinyLong", 128, 80, 2, 128, 4, 77, 49408, 128, 2, 2)

CONFIGS = {c.name: c for c in (VIT_B16, VIT_L14, VIT_L14_336, VIT_B32, TINY, TINY14, TINY_LONG)}


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(key=[seed & 0xFFFFFFFFFFFFFFFF, zlib.crc32(name.encode())]))


def normal(seed: int, name: str, shape, std: float = 1.0, mean: float = 0.0) -> np.ndarray:
    x = _rng(seed, name).standard_normal(size=shape, dtype=np.float32)
    if std != 1.0:
        x *= np.float32(std)
    if mean != 0.0:
        x += np.float32(mean)
    return x


def _block(sd, seed, prefix, width, layers):
    attn_std = width ** -0.5
    proj_std = (width ** -0.5) * ((2 * layers) ** -0.5)
    fc_std = (2 * width) ** -0.5
    for i in range(layers):
        p = f"{prefix}resblocks.{i}."
        for nm, shape, std, mean in (
            ("attn.in_proj_weight", (3 * width, width), attn_std, 0.0),
            ("attn.in_proj_bias", (3 * width,), 0.02, 0.0),
            ("attn.out_proj.weight", (width, width), proj_std, 0.0),
            ("attn.out_proj.bias", (width,), 0.02, 0.0),
            ("ln_1.weight", (width,), 0.1, 1.0),
            ("ln_1.bias", (width,), 0.05, 0.0),
            ("mlp.c_fc.weight", (4 * width, width), fc_std, 0.0),
            ("mlp.c_fc.bias", (4 * width,), 0.02, 0.0),
            ("mlp.c_proj.weight", (width, 4 * width), proj_std, 0.0),
            ("mlp.c_proj.bias", (width,), 0.02, 0.0),
            ("ln_2.weight", (width,), 0.1, 1.0),
            ("ln_2.bias", (width,), 0.05, 0.0),
        ):
            sd[p + nm] = normal(seed, p + nm, shape, std, mean)


def clip_state_dict(cfg: ClipConfig, seed: int = WEIGHT_SEED, outliers: str | None = None) -> dict:
    """Synthetic state dict with the reference's key names / shapes (``model.py:418-441``).

    outliers (round 6; the statistics real CLIP checkpoints show and N(0, sigma) weights do not): ``'channels'`` — "massive activation" channels: the first
    block's ``c_proj`` bias drives two residual channels of each tower to +150 / -90 (100-300x the stream's typical magnitude of 0.5-1, carried by the
    residual connection through every later block) and one ``ln_pre`` / first ``ln_1`` gain is 40; ``'offset'`` — every row of the stream gets a common offset
    of about 12 standard deviations (``ln_pre`` bias; the text tower's first ``c_proj`` bias), the case where one-sweep variance E[x^2] - mean^2 loses digits and the
    engine's guard (mean^2 > 64 var) must switch to two-sweep statistics."""
    sd = _clip_state_dict(cfg, seed)
    if outliers is None:
        return sd
    vw, tw = cfg.vision_width, cfg.transformer_width
    if outliers == "channels":
        for pre, w in (("visual.transformer.", vw), ("transformer.", tw)):
            b = sd[pre + "resblocks.0.mlp.c_proj.bias"]
            b[37 % w] += np.float32(150.0)
            b[(w * 5) // 8 + 3] -= np.float32(90.0)
        sd["visual.ln_pre.weight"][11] = np.float32(40.0)
        sd["transformer.resblocks.0.ln_1.weight"][11] = np.float32(40.0)
    elif outliers == "offset":
        sd["visual.ln_pre.bias"] += np.float32(12.0)
        # the text stream starts at a deviation of 0.02 and has 0.2-0.9 behind the first block: the offset enters there (an offset of 6 on the embeddings
        # themselves would put them below fp16's resolution — in the reference's fp16 as much as in any other)
        sd["transformer.resblocks.0.mlp.c_proj.bias"] += np.float32(6.0)
    else:
        raise ValueError(outliers)
    return sd


def _clip_state_dict(cfg: ClipConfig, seed: int) -> dict:
    sd = {}
    vw, tw = cfg.vision_width, cfg.transformer_width
    ps = cfg.vision_patch_size
    scale = vw ** -0.5
    sd["visual.class_embedding"] = normal(seed, "visual.class_embedding", (vw,), scale)
    sd["visual.positional_embedding"] = normal(seed, "visual.positional_embedding", (cfg.n_patches + 1, vw), scale)
    sd["visual.proj"] = normal(seed, "visual.proj", (vw, cfg.embed_dim), scale)
    sd["visual.conv1.weight"] = normal(seed, "visual.conv1.weight", (vw, 3, ps, ps), (3 * ps * ps) ** -0.5)
    for nm in ("ln_pre", "ln_post"):
        sd[f"visual.{nm}.weight"] = normal(seed, f"visual.{nm}.weight", (vw,), 0.1, 1.0)
        sd[f"visual.{nm}.bias"] = normal(seed, f"visual.{nm}.bias", (vw,), 0.05)
    _block(sd, seed, "visual.transformer.", vw, cfg.vision_layers)
    _block(sd, seed, "transformer.", tw, cfg.transformer_layers)
    sd["token_embedding.weight"] = normal(seed, "token_embedding.weight", (cfg.vocab_size, tw), 0.02)
    sd["positional_embedding"] = normal(seed, "positional_embedding", (cfg.context_length, tw), 0.01)
    sd["ln_final.weight"] = normal(seed, "ln_final.weight", (tw,), 0.1, 1.0)
    sd["ln_final.bias"] = normal(seed, "ln_final.bias", (tw,), 0.05)
    sd["text_projection"] = normal(seed, "text_projection", (tw, cfg.embed_dim), tw ** -0.5)
    sd["logit_scale"] = np.array(np.log(1 / 0.07), dtype=np.float32)
    return sd


PROMPT_NAMES = ("dim_1_share", "dim_2_visual", "dim_2_textual", "dim_3_visual", "dim_3_textual")


def prompt_factors(layer_num: int, prompt_num: int, vis_dim: int, txt_dim: int, r: int = 4,
                   seed: int = PROMPT_SEED, task: int = 0) -> dict:
    """Five CP factors ~ N(0, 0.5^2), the reference's init (``prompts.py:21-25``)."""
    shapes = {
        "dim_1_share": (layer_num, r),
        "dim_2_visual": (prompt_num, r),
        "dim_2_textual": (prompt_num, r),
        "dim_3_visual": (vis_dim, r),
        "dim_3_textual": (txt_dim, r),
    }
    return {k: normal(seed, f"prompts.{task}.{k}", s, 0.5) for k, s in shapes.items()}


def task_family_factors(family: str, task: int, vis_dim: int, txt_dim: int, r: int = 4, layer_num: int = 9, prompt_num: int = 16) -> dict:
    """The factors of task `task` in a 12-task session of the task-loss fixtures (tests/golden/task_loss_wide.npz): 'random' = independent draws at the
    reference's init scale (prompt_factors(task=t)); 'drift' = a common ancestor (task 0's draw) plus a step of 2 % of the scale per task in an independent
    direction — neighbouring tasks stay nearly parallel (cosine of the flattened stacks close to one)."""
    base = prompt_factors(layer_num, prompt_num, vis_dim, txt_dim, r=r, task=task)
    if family == "mixed":      # tasks 0-3 drift around the ancestor, the later ones are independent: saturated and live pairs in one matrix
        family = "drift" if task < 4 else "random"
    if family == "random":
        return base
    if family != "drift":
        raise ValueError(family)
    anc = prompt_factors(layer_num, prompt_num, vis_dim, txt_dim, r=r, task=0)
    return {k: (anc[k] + np.float32(0.02 * task) * base[k]).astype(np.float32) for k in anc}


def images(batch: int, resolution: int, seed: int = IMAGE_SEED) -> np.ndarray:
    """N(0,1) stand-in for ImageNet-normalised pixels (``utils/data.py:310-313``)."""
    return normal(seed, "images", (batch, 3, resolution, resolution))


def token_ids(batch: int, context_length: int = 77, n_ctx: int = 16, seed: int = TOKEN_SEED,
              min_len: int = 5, max_len: int = 40) -> np.ndarray:
    """Token ids shaped like ``PromptLearner.forward`` builds them (``prompt_learner.py:128-133``):
    SOT, n_ctx placeholder "X" tokens, caption ids, '.', EOT, zero padding."""
    g = _rng(seed, "token_ids")
    ids = np.zeros((batch, context_length), dtype=np.int64)
    max_len = min(max_len, context_length - n_ctx - 3)
    for b in range(batch):
        n = int(g.integers(min_len, max_len + 1))
        body = g.integers(320, 49405, size=n)
        row = [SOT] + [X_TOKEN] * n_ctx + body.tolist() + [DOT_TOKEN, EOT]
        ids[b, :len(row)] = row
    return ids


INTERACT_SEED = 4000


def interact_inputs(bs: int = 2, P: int = 16, Dv: int = 96, Dt: int = 768, seed: int = INTERACT_SEED):
    """Inputs of the InteractModule fixture (tests/golden/interact.npz): visual / textual prompt rows and the weights of the scalar test loss
    sum(v_out * wv) + sum(t_out * wt), regenerated from the seed on both sides instead of being stored."""
    return {k: normal(seed, "interact." + k, (bs, P, D)) for k, D in (("visual_in", Dv), ("textual_in", Dt), ("wv", Dv), ("wt", Dt))}


KMEANS_SEED = 20240707


def clustering_features(n: int = 600, dim: int = 512, modes: int = 7, seed: int = KMEANS_SEED):
    """Un-normalised synthetic features for the task-key clustering (methods/sprompt.py:370-397): a mixture of `modes` Gaussian blobs of unequal size and
    spread around random directions — more blobs than the five KMeans centres, so that the fit has real decisions to make — regenerated from the seed on
    both sides (the fixture tests/golden/kmeans.npz holds only the centres the imported reference found).  -> (visual [n, dim], textual [n, dim]) f32."""
    out = []
    for name in ("visual", "textual"):
        g = _rng(seed, "kmeans." + name)
        centres = g.standard_normal((modes, dim)) * 1.5
        sizes = g.dirichlet(np.full(modes, 2.0))
        which = g.choice(modes, size=n, p=sizes)
        spread = 0.6 + 0.8 * g.random(modes)
        x = centres[which] + g.standard_normal((n, dim)) * spread[which][:, None]
        out.append(x.astype(np.float32))
    return out[0], out[1]


def duplicate_heavy_features(n: int = 24, distinct: int = 4, dim: int = 4, seed: int = 1):
    """L2-normalised features with FEWER distinct rows than the 5 clusters of the task keys (a tiny task whose images repeat: COCO pairs every image with
    ~5 captions, so a task's image features come in runs of equal rows).  k-means++ then has to seed one centre on a row it already chose, that cluster
    receives no point in the Lloyd iteration, and scikit-learn's empty-cluster relocation (_relocate_empty_clusters_dense) decides how the fit goes on —
    where the first builds of lpi_amd.kmeans raised.  (Recipe chosen so that numpy's tie order and the device's agree with scikit-learn's.)"""
    g = np.random.default_rng(seed)
    g.integers(10, 40), g.integers(2, 6), g.integers(2, 5)          # (keeps the stream position of the search that found the recipe)
    base = g.standard_normal((distinct, dim)).astype(np.float32)
    x = base[g.integers(0, distinct, size=n)]
    return (x / np.linalg.norm(x, axis=-1, keepdims=True)).astype(np.float32)

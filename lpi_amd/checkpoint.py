"""CLIP checkpoints as the reference consumes them, and the frozen backbone as module state.

Reference: ``load_clip_to_cpu`` (models/clip/prompt_learner.py:10-40) downloads OpenAI's file, tries ``torch.jit.load(path).state_dict()`` (the
published files are TorchScript archives) and falls back to ``torch.load``; ``build_model`` (models/clip/model.py:418-524) then takes EVERY width /
layer count / patch size from the tensors' shapes, drops the three scalar entries ``input_resolution / context_length / vocab_size`` (:443-445),
converts to fp16 (``convert_weights``, :394-415) and loads.  There is no network here, so the download is the caller's business
(``args['clip_state_dict']`` = a state dict or a path); everything after it is restated in this file:

  * ``load_clip_state_dict``  dict | nn.Module | path (TorchScript archive, or a torch.save'd state dict / module) -> {name: f32 CPU tensor};
  * ``infer_config``           the shape inference of build_model, ViT only (RN50-style checkpoints are refused loudly: the hot path is the ViT);
  * ``ParamTree``              an nn.Module tree whose ``state_dict()`` keys are exactly the given dotted names — how SliNet registers the frozen CLIP
                               tensors, so that ``count_parameters(model._network)`` (trainer.py:50-51) sees 149.78 M like the reference's network.

The engine keeps its own operand copies (bf16 / fp16 / f32, some transposed); fp16 checkpoints are widened to f32 first (exact), so a checkpoint that
went through ``convert_weights`` gives the same operands as its f32 original wherever the mode rounds to fp16 / bf16 anyway.
"""
from __future__ import annotations

import os

import numpy as np
import torch
from torch import nn

from .synth import ClipConfig

_SCALARS = ("input_resolution", "context_length", "vocab_size")      # model.py:443-445


class CheckpointError(ValueError):
    pass


def _as_tensor(v) -> torch.Tensor:
    t = v if torch.is_tensor(v) else torch.as_tensor(np.asarray(v))
    t = t.detach()
    if t.is_floating_point():
        t = t.to(device="cpu", dtype=torch.float32)
    return t


def _from_file(path: str, trusted: bool):
    """prompt_learner.py:15-22: a TorchScript archive first, else a pickled object."""
    try:
        return torch.jit.load(path, map_location="cpu").eval().state_dict()
    except RuntimeError:
        pass
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except Exception as e:      # noqa: BLE001 — a pickled nn.Module (or anything else torch's safe loader refuses)
        if not trusted:
            raise CheckpointError(
                f"{path} is neither a TorchScript archive nor a plain tensor state dict; loading it needs a full unpickle (arbitrary code). "
                "Pass args['clip_checkpoint_trusted'] = True if the file is yours.") from e
        return torch.load(path, map_location="cpu", weights_only=False)


def load_clip_state_dict(src, trusted: bool = False) -> dict:
    """-> {name: CPU tensor (floating point entries as f32)} without the three scalar entries; ``src`` as described in the module docstring."""
    if isinstance(src, (str, os.PathLike)):
        if not os.path.isfile(src):
            raise FileNotFoundError(f"CLIP checkpoint {src} not found")
        src = _from_file(os.fspath(src), trusted)
    if hasattr(src, "state_dict") and callable(src.state_dict):
        src = src.state_dict()
    if isinstance(src, dict) and "state_dict" in src and isinstance(src["state_dict"], dict) and "text_projection" not in src:
        src = src["state_dict"]
    if not isinstance(src, dict):
        raise CheckpointError(f"cannot take CLIP weights from a {type(src).__name__}")
    sd = {k: _as_tensor(v) for k, v in src.items() if k not in _SCALARS}
    return sd


def infer_config(sd: dict, name: str = "checkpoint") -> ClipConfig:
    """build_model's shape inference (model.py:419-441).  Raises CheckpointError for ModifiedResNet checkpoints (no ``visual.proj``) and for tensors
    the ViT path needs but the dict lacks."""
    if "visual.proj" not in sd:
        rn = any(k.startswith("visual.layer1") for k in sd)
        raise CheckpointError(("this is a ModifiedResNet (RN50-style) CLIP checkpoint: only the ViT backbones of the LPI configs are built"
                               if rn else "not a CLIP state dict: 'visual.proj' is missing"))
    need = ("visual.conv1.weight", "visual.positional_embedding", "text_projection", "positional_embedding", "token_embedding.weight",
            "ln_final.weight", "visual.class_embedding", "logit_scale")
    missing = [k for k in need if k not in sd]
    if missing:
        raise CheckpointError(f"CLIP state dict lacks {missing}")
    shape = lambda k: tuple(sd[k].shape)  # noqa: E731
    vision_width = shape("visual.conv1.weight")[0]
    vision_layers = len([k for k in sd if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
    vision_patch_size = shape("visual.conv1.weight")[-1]
    grid = round((shape("visual.positional_embedding")[0] - 1) ** 0.5)
    if grid * grid + 1 != shape("visual.positional_embedding")[0]:
        raise CheckpointError("visual.positional_embedding does not hold 1 + grid^2 rows")
    embed_dim = shape("text_projection")[1]
    context_length = shape("positional_embedding")[0]
    vocab_size = shape("token_embedding.weight")[0]
    transformer_width = shape("ln_final.weight")[0]
    transformer_heads = transformer_width // 64
    transformer_layers = len(set(k.split(".")[2] for k in sd if k.startswith("transformer.resblocks")))
    return ClipConfig(name, embed_dim, vision_patch_size * grid, vision_layers, vision_width, vision_patch_size, context_length, vocab_size,
                      transformer_width, transformer_heads, transformer_layers)


def same_architecture(a: ClipConfig, b: ClipConfig) -> bool:
    return a.as_clip_args() == b.as_clip_args()


class ParamTree(nn.Module):
    """Frozen tensors as a module tree: ``ParamTree({'visual.conv1.weight': t, ...}).state_dict()`` has exactly those keys, in that order."""

    def __init__(self, tensors: dict | None = None):
        super().__init__()
        for name, t in (tensors or {}).items():
            node = self
            *parents, leaf = name.split(".")
            for p in parents:
                if p not in node._modules:
                    node.add_module(p, ParamTree())
                node = node._modules[p]
            t = (t if torch.is_tensor(t) else torch.as_tensor(np.asarray(t))).detach()      # dtype and device as given
            node.register_parameter(leaf, nn.Parameter(t, requires_grad=False)) if t.is_floating_point() else node.register_buffer(leaf, t)

    def forward(self):          # a container: never called (present so that torch.jit.script accepts the tree in the checkpoint tests)
        return None

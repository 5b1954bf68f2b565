"""The CLIP checkpoint loader on CPU (lpi_amd/checkpoint.py; the reference's load_clip_to_cpu + build_model, models/clip/prompt_learner.py:10-40 and
models/clip/model.py:418-524): a ViT-B/16-shaped fp16 checkpoint written BOTH ways the reference can meet it — a TorchScript archive (OpenAI's
published format: torch.jit.load(...).state_dict()) and a torch.save'd state dict — comes back as f32 tensors without the three scalar entries, and the
architecture inferred from the shapes is synth.VIT_B16; ModifiedResNet checkpoints and pickled objects are refused loudly; and the host-side batch
gather of the input pipeline (lpi_host_gather: no GPU involved)."""
import ctypes

import numpy as np
import pytest
import torch

from lpi_amd import synth
from lpi_amd.checkpoint import CheckpointError, ParamTree, infer_config, load_clip_state_dict, same_architecture


@pytest.fixture(scope="module")
def vitb16_half():
    sd = {k: torch.as_tensor(np.asarray(v)).half() for k, v in synth.clip_state_dict(synth.VIT_B16).items()}
    sd.update(input_resolution=torch.tensor(224), context_length=torch.tensor(77), vocab_size=torch.tensor(49408))      # as in OpenAI's archives
    return sd


def test_vitb16_fp16_checkpoint_both_ways(vitb16_half, tmp_path):
    p_dict, p_jit = str(tmp_path / "sd.pt"), str(tmp_path / "jit.pt")
    torch.save(vitb16_half, p_dict)
    torch.jit.save(torch.jit.script(ParamTree(vitb16_half)), p_jit)
    assert torch.jit.load(p_jit).state_dict()["visual.proj"].dtype == torch.float16          # the archive holds fp16, like convert_weights' output
    for src in (p_dict, p_jit, vitb16_half, ParamTree(vitb16_half)):
        sd = load_clip_state_dict(src)
        assert not ({"input_resolution", "context_length", "vocab_size"} & set(sd))
        assert all(v.dtype == torch.float32 and v.device.type == "cpu" for v in sd.values())
        assert len(sd) == len(vitb16_half) - 3 and sum(v.numel() for v in sd.values()) == 149_620_737
        cfg = infer_config(sd, "ViT-B/16")
        assert cfg == synth.VIT_B16 and same_architecture(cfg, synth.VIT_B16)
        w = "transformer.resblocks.7.mlp.c_fc.weight"
        assert torch.equal(sd[w], vitb16_half[w].float())          # widened exactly


def test_slinet_takes_the_architecture_from_the_checkpoint(vitb16_half, tmp_path):
    import json
    import os
    from lpi_amd.retrieval.models.slinet import SliNet
    args = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lpi_amd", "retrieval", "configs", "lpi", "coco_lpi.json")))
    args["device"] = [torch.device("cpu")]
    path = str(tmp_path / "ViT-B-16.pt")
    torch.save(vitb16_half, path)
    net = SliNet(dict(args, clip_state_dict=path))
    assert net.clip_cfg == synth.VIT_B16
    assert torch.equal(net.clip_model.visual.conv1.weight, vitb16_half["visual.conv1.weight"].float())
    # a checkpoint of another architecture under this name is an error, not a silent mix of shapes
    tiny = {k: torch.as_tensor(np.asarray(v)) for k, v in synth.clip_state_dict(synth.TINY).items()}
    with pytest.raises(ValueError, match="backbonename"):
        SliNet(dict(args, clip_state_dict=tiny))
    # ... and an unknown name takes whatever the file holds (build_model never looks at the name, model.py:419-441)
    net = SliNet(dict(args, clip_state_dict=tiny, backbonename="my-clip", visual_dim=128, textual_dim=128))
    assert same_architecture(net.clip_cfg, synth.TINY) and net.clip_cfg.name == "my-clip"


def test_refused_checkpoints(tmp_path):
    rn = {"visual.layer1.0.conv1.weight": torch.zeros(64, 64, 1, 1), "text_projection": torch.zeros(512, 1024)}
    with pytest.raises(CheckpointError, match="ModifiedResNet"):
        infer_config(load_clip_state_dict(rn))
    with pytest.raises(CheckpointError, match="visual.proj"):
        infer_config({"a": torch.zeros(1)})
    with pytest.raises(FileNotFoundError):
        load_clip_state_dict(str(tmp_path / "missing.pt"))
    p = str(tmp_path / "module.pt")
    torch.save(ParamTree({"visual.proj": torch.zeros(2, 2)}), p)          # a pickled nn.Module: needs a full unpickle
    with pytest.raises(CheckpointError, match="trusted"):
        load_clip_state_dict(p)
    assert "visual.proj" in load_clip_state_dict(p, trusted=True)


def test_host_gather_copies_rows_on_several_threads():
    from lpi_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(0)
    rows = [torch.randn(3, 37, 41, generator=g) for _ in range(9)]
    rows.append(rows[2])                                                    # an aliased source (a dataset that repeats an image)
    each = rows[0].numel() * 4
    for threads in (0, 1, 4, 64):
        dst = torch.full((len(rows), 3, 37, 41), float("nan"))
        ptrs = (ctypes.c_void_p * len(rows))(*[r.data_ptr() for r in rows])
        assert lib.lpi_host_gather(dst.data_ptr(), ctypes.cast(ptrs, ctypes.c_void_p), len(rows), each, threads) == 0
        assert torch.equal(dst, torch.stack(rows))
    big = torch.randn(2, 3 * (1 << 20) // 4 + 5, generator=g)             # rows longer than the 1 MiB work unit, not a multiple of it
    dst = torch.empty_like(big)
    ptrs = (ctypes.c_void_p * 2)(big[0].data_ptr(), big[1].data_ptr())
    assert lib.lpi_host_gather(dst.data_ptr(), ctypes.cast(ptrs, ctypes.c_void_p), 2, big.shape[1] * 4, 8) == 0 and torch.equal(dst, big)
    assert lib.lpi_host_gather(None, ctypes.cast(ptrs, ctypes.c_void_p), 2, 16, 1) == -22
    assert lib.lpi_host_gather(dst.data_ptr(), ctypes.cast(ptrs, ctypes.c_void_p), 0, 16, 1) == -22

"""Host logic of the plugin surface on CPU: construction through the reference's factory entry, the parameter naming /
trainable-filter contract (sprompt.py:230-237), config handling, loud failure without a GPU, and that the REFERENCE's own
trainer.py / main.py resolve to this implementation when lpi_amd/retrieval is first on sys.path."""
import json
import os
import subprocess
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RET = os.path.join(REPO, "lpi_amd", "retrieval")


def load_args(**over):
    args = json.load(open(os.path.join(RET, "configs", "lpi", "coco_lpi.json")))
    args["device"] = [torch.device("cpu")]
    args.update(over)
    return args


def test_factory_builds_sprompts_with_reference_contract():
    from lpi_amd.retrieval.utils import factory
    m = factory.get_model("sprompts", load_args())
    net = m._network
    names = [n for n, _ in net.named_parameters()]
    assert len([n for n in names if n.startswith("prompts.")]) == 12 * 5
    # the reference network's state (trainer.py:50: 'All params' 149.78 M, SURVEY section 6 probe): CLIP ViT-B/16 (149 620 737, registered frozen under
    # clip_model.*) + 12 DecomposedPrompts (12 x 5 284) + the 12 PromptLearners' unused ctx vectors (12 x 16 x 512)
    from lpi_amd.retrieval.utils.toolkit import count_parameters
    assert count_parameters(net) == 149_620_737 + 12 * 5284 + 12 * 16 * 512 == 149_782_449
    assert "clip_model.visual.conv1.weight" in names and "clip_model.transformer.resblocks.11.mlp.c_proj.bias" in names and "classifier_pool.3.ctx" in names
    assert not any(p.requires_grad for n, p in net.named_parameters() if not n.startswith("prompts."))
    for t in range(12):
        for k in ("dim_1_share", "dim_2_visual", "dim_2_textual", "dim_3_visual", "dim_3_textual"):
            assert f"prompts.{t}.{k}" in names
    net.update_fc(0)
    assert net.numtask == 1
    for name, p in net.named_parameters():                  # the reference's substring filter
        p.requires_grad_("prompts." + str(net.numtask - 1) + "." in name)
    train = [p for p in net.parameters() if p.requires_grad]
    assert len(train) == 5 and sum(p.numel() for p in train) == 5284
    assert net.feature_dim == 512 and net.class_num == 2 and net.depth == 1
    for attr in ("forward", "cal_loss", "extract_vector", "extract_textual_vector", "visual_interface", "textual_interface",
                 "update_fc", "copy", "freeze"):
        assert callable(getattr(net, attr))
    for attr in ("incremental_train", "after_task"):
        assert callable(getattr(m, attr))
    with pytest.raises(KeyError):
        factory.get_model("nope", load_args())


def test_unknown_net_and_prompt_type_raise():
    from lpi_amd.retrieval.methods.sprompt import SPrompts
    with pytest.raises(ValueError):
        SPrompts(load_args(net_type="sip"))
    with pytest.raises(ValueError):
        SPrompts(load_args(prompt_type="l2p"))


def test_prompt_depth_semantics():
    from lpi_amd.retrieval.models.slinet import SliNet
    assert SliNet(load_args()).depth == 1                                  # shipped reference: depth is dead (F1)
    assert SliNet(load_args(honor_prompt_depth=True)).depth == 3


def test_no_cpu_fallback():
    from lpi_amd import _lib
    from lpi_amd.retrieval.models.slinet import SliNet
    net = SliNet(load_args(backbonename="tiny", visual_dim=128, textual_dim=128))
    net.update_fc(0)
    with pytest.raises(_lib.LpiError):
        net(torch.zeros(2, 3, 32, 32), torch.zeros(2, 77, dtype=torch.long))


def test_copy_shares_engine_and_freeze():
    from lpi_amd.retrieval.models.slinet import SliNet
    net = SliNet(load_args(backbonename="tiny", visual_dim=128, textual_dim=128))
    net.engine = object()
    c = net.copy().freeze()
    assert c.engine is net.engine and c is not net
    assert all(not p.requires_grad for p in c.parameters())
    assert c.prompts[0].dim_1_share.data_ptr() != net.prompts[0].dim_1_share.data_ptr()
    # the frozen f32 masters are shared like the engine (the reference deep-copies 149.78 M parameters after every task)
    assert c.clip_model is net.clip_model and "clip_model" in dict(net.named_children())


def test_full_state_dict_round_trip_drops_the_stale_engine():
    """SliNet.state_dict() holds what the reference network's does — backbone, prompts, the PromptLearners' ctx — and loads into a fresh network; operand
    copies built from the old backbone are dropped."""
    from lpi_amd.retrieval.models.slinet import SliNet
    a = SliNet(load_args(backbonename="tiny", visual_dim=128, textual_dim=128))
    b = SliNet(load_args(backbonename="tiny", visual_dim=128, textual_dim=128))
    sd = a.state_dict()
    assert {k.split(".")[0] for k in sd} == {"prompts", "classifier_pool", "clip_model"}
    b.engine = object()
    b.load_state_dict(sd)
    assert b.engine is None
    for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.equal(p, q), n
    b.engine = object()
    b.load_state_dict({k: v for k, v in sd.items() if k.startswith("prompts.")}, strict=False)      # prompts only: the engine stays
    assert b.engine is not None


def test_learner_state_round_trip():
    """the state a continual run carries besides the frozen backbone — 12 x 5 prompt factors, numtask, the KMeans task keys of both
    modalities — through SPrompts.state_dict / load_state_dict (and torch.save / torch.load of it)."""
    import io
    from lpi_amd.retrieval.methods.sprompt import SPrompts
    a = SPrompts(load_args(backbonename="tiny", visual_dim=128, textual_dim=128))
    a._network.numtask, a.cur_id = 3, 2
    a.all_keys = [torch.randn(5, 128) for _ in range(3)]
    a.textual_all_keys = [torch.randn(5, 128) for _ in range(3)]
    buf = io.BytesIO()
    torch.save(a.state_dict(), buf)
    buf.seek(0)
    b = SPrompts(load_args(backbonename="tiny", visual_dim=128, textual_dim=128))
    assert not torch.equal(b._network.prompts[2].dim_3_visual, a._network.prompts[2].dim_3_visual)
    b.load_state_dict(torch.load(buf, weights_only=True))
    assert b._network.numtask == 3 and b.cur_id == 2 and len(b.all_keys) == 3
    for (n, p), (_, q) in zip(a._network.named_parameters(), b._network.named_parameters()):
        if n.startswith("prompts."):
            assert torch.equal(p, q), n
    assert all(torch.equal(x, y) for x, y in zip(a.textual_all_keys, b.textual_all_keys))
    with pytest.raises(KeyError):
        b._network.load_trainable_state_dict({"clip_model.logit_scale": torch.zeros(())})


def test_pre_caption_and_synthetic_dataset_contract():
    from lpi_amd.retrieval.utils.data import SyntheticCoco, SyntheticCocoEval, pre_caption
    assert pre_caption("A man, riding a   wave-board!", 50) == "a man riding a wave board"
    img, cap, z, task = SyntheticCoco(4, [3], 32)[1]
    assert img.shape == (3, 32, 32) and cap.shape == (77,) and z == 0 and task == 3
    # the reference's item structure (utils/data.py:376-382): f32 image, caption STRING, 0, task; pooled images are views (no per-item generation)
    ds = SyntheticCoco(6, [2], 32, captions="strings", image_pool=4)
    img, cap, z, task = ds[5]
    assert img.dtype == torch.float32 and isinstance(cap, str) and 1 <= len(cap.replace(" ", "")) <= 40 and task == 2
    assert ds[5][0].data_ptr() == ds[1][0].data_ptr() and not torch.equal(ds[0][0], ds[1][0])
    from lpi_amd.retrieval.utils.data import collate_keep_images
    imgs, caps, zs, tasks = collate_keep_images([ds[i] for i in range(3)])
    assert isinstance(imgs, list) and len(imgs) == 3 and caps == [ds[i][1] for i in range(3)] and tasks.tolist() == [2, 2, 2]
    ev = SyntheticCocoEval(3, [0, 1], cpi=2, resolution=32)
    assert len(ev) == 6 and len(ev.text) == 12 and ev.txt2img[5] == 2 and ev.img2txt[2] == [4, 5]


@pytest.mark.skipif(not os.path.isdir("/root/reference/retrieval"), reason="reference checkout not present")
def test_reference_trainer_runs_unchanged_against_this_plugin():
    """Import the reference's own trainer.py with lpi_amd/retrieval first on sys.path: its ``from utils import factory``
    must resolve to this implementation and build the HIP-backed learner from the reference's flat args dict."""
    code = f"""
import sys, json, importlib.util, torch
sys.path.insert(0, {RET!r})
spec = importlib.util.spec_from_file_location('ref_trainer', '/root/reference/retrieval/trainer.py')
tr = importlib.util.module_from_spec(spec); spec.loader.exec_module(tr)
args = json.load(open('/root/reference/retrieval/configs/lpi/coco_lpi.json'))
args['device'] = [torch.device('cpu')]
m = tr.factory.get_model(args['model_name'], args)
print(type(m).__module__, type(m._network).__module__, tr.count_parameters(m._network))
"""
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=RET)
    assert out.returncode == 0, out.stderr
    # 'All params' as the reference's trainer.py:50 logs it for its own network: 149.78 M (SURVEY section 6 probe)
    assert "lpi_amd.retrieval.methods.sprompt lpi_amd.retrieval.models.slinet 149782449" in out.stdout

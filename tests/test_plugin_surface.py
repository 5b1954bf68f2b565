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
    assert len(names) == 12 * 5
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


def test_pre_caption_and_synthetic_dataset_contract():
    from lpi_amd.retrieval.utils.data import SyntheticCoco, SyntheticCocoEval, pre_caption
    assert pre_caption("A man, riding a   wave-board!", 50) == "a man riding a wave board"
    img, cap, z, task = SyntheticCoco(4, [3], 32)[1]
    assert img.shape == (3, 32, 32) and cap.shape == (77,) and z == 0 and task == 3
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
    assert "lpi_amd.retrieval.methods.sprompt lpi_amd.retrieval.models.slinet 63408" in out.stdout

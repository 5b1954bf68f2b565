"""Round-6 additions on a real MI355X:
  * the task loss (slinet.py:167-183 -> loss/loss.py:6-33) at the sizes the reference runs it — stacks of [t+1, 110 592] / [t+1, 73 728] for 2, 7 and 12 tasks at
    temperature 0.001 — against fixtures made by the imported `SliNet.cal_task_loss` (tests/golden/task_loss_wide.npz): through the plugin's `cal_task_loss` (autograd
    over lpi_nt_bxent_fwd_bwd + the CP backward) and through the raw C entry point;
  * the whole step of task 12 of a continual session (ViT-B/16, 8 pairs) against the reference's — both the reference-ordered calls (net -> cal_loss -> backward) and
    the fused `SliNet.train_step` whose task term adds onto the seeded prompt-gradient buffers;
  * restoring a checkpoint drops the fused task term's cached rows."""
import json
import os
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from lpi_amd import _lib, synth  # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RET = os.path.join(REPO, "lpi_amd", "retrieval")
DEV = torch.device("cuda:0")
CFG = synth.VIT_B16


def rel_err(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


# ------------------------------------------------------------------------------------------------ a9 at operating size
@pytest.mark.parametrize("fam", ["random", "drift", "mixed"])
@pytest.mark.parametrize("numtask", [2, 7, 12])
def test_task_loss_at_operating_size_matches_reference(golden, fam, numtask):
    from lpi_amd.retrieval.models.prompts.prompts import DecomposedPrompt
    from lpi_amd.retrieval.models.slinet import SliNet
    g = golden("task_loss_wide")
    mods = []
    for t in range(numtask):
        m = DecomposedPrompt(9, 16, CFG.vision_width, CFG.transformer_width).to(DEV)
        for k, v in synth.task_family_factors(fam, t, CFG.vision_width, CFG.transformer_width).items():
            getattr(m, k).data = torch.from_numpy(v.copy()).to(DEV)
        for p in m.parameters():
            p.requires_grad_(t == numtask - 1)
        mods.append(m)
    # the plugin's own method, unbound, on a stand-in that holds what it reads (exactly how the fixture was made from the reference's method)
    ns = types.SimpleNamespace(prompts=mods, _task_target=SliNet._task_target)
    n0 = _lib.launch_count()
    loss = SliNet.cal_task_loss(ns, numtask - 1, None, None)
    loss.backward()
    torch.cuda.synchronize()
    assert _lib.launch_count() - n0 >= 4
    ref = float(g[f"{fam}.{numtask}.loss"])
    assert abs(float(loss.detach()) - ref) <= 1e-4 * max(1.0, abs(ref)), (float(loss.detach()), ref)
    worst = 0.0
    for k in synth.PROMPT_NAMES:
        r = g[f"{fam}.{numtask}.grad.{k}"]
        got = getattr(mods[-1], k).grad.cpu().numpy()
        if np.abs(r).max() == 0.0:
            assert np.abs(got).max() <= 1e-12, k      # saturated inner sigmoid: the reference's gradient is exactly zero
        else:
            worst = max(worst, rel_err(got, r))
            assert rel_err(got, r) <= 5e-3, (k, rel_err(got, r))
    print(f"task loss, {fam}, {numtask} tasks: {float(loss.detach()):.7f} (reference {ref:.7f}); factor gradients {worst:.2e} relative")


@pytest.mark.parametrize("numtask", [7, 12])
def test_nt_bxent_entry_point_at_operating_size_against_f64(numtask):
    """lpi_nt_bxent_fwd_bwd itself on the visual stack of the 'random' family ([T, 110 592], cosines of a few 1e-3, temperature 0.001: cos / T of order one to
    ten) against f64 autograd of loss.py:6-33 as written — the loss and the dense gradient of the current task's row."""
    from lpi_amd.engine import call, _stream
    D = 9 * 16 * CFG.vision_width
    rows = []
    for t in range(numtask):
        f = {k: torch.from_numpy(v).double() for k, v in synth.prompt_factors(9, 16, CFG.vision_width, CFG.transformer_width, task=t).items()}
        rows.append(torch.einsum("lr,pr,dr->lpd", f["dim_1_share"], f["dim_2_visual"], f["dim_3_visual"]).reshape(-1) / 4)
    X = torch.stack(rows)
    sim = np.loadtxt(os.path.join(RET, "MID", "task_sim_matrix.txt"))[:numtask, :numtask]
    tgt = torch.from_numpy((sim > 0.4).astype(np.int32))
    Xr = X.clone().requires_grad_(True)
    xn = Xr / Xr.norm(dim=-1, keepdim=True)
    cs = (xn @ xn.t()).masked_fill(torch.eye(numtask, dtype=torch.bool), float("inf"))
    l = torch.nn.functional.binary_cross_entropy_with_logits((cs / 0.001).sigmoid(), tgt.double(), reduction="none")
    pos = tgt.bool()
    ref = ((l * pos).sum(1) / pos.sum(1) + (l * ~pos).sum(1) / (~pos).sum(1)).mean()
    ref.backward()
    row = numtask - 1
    loss, dx, scratch = torch.zeros(1, device=DEV), torch.zeros(D, device=DEV), torch.zeros(2 * numtask * numtask, device=DEV)
    call("lpi_nt_bxent_fwd_bwd", numtask, D, row, X.float().to(DEV).contiguous(), tgt.to(DEV), 0.001, 1.0, loss, dx, 0, scratch, _stream())
    torch.cuda.synchronize()
    assert abs(loss.item() - ref.item()) < 2e-5 * max(1.0, abs(ref.item()))
    gref = Xr.grad[row]
    assert float(gref.abs().max()) > 0
    # f32 cosines carry ~1e-7 of error, i.e. 1e-4 of the sigmoid's argument at this temperature
    assert float((dx.double().cpu() - gref).abs().max()) <= 3e-3 * float(gref.abs().max())


# ------------------------------------------------------------------------------------------------ task 12's whole step
def vitb16_args(**over):
    args = json.load(open(os.path.join(RET, "configs", "lpi", "coco_lpi.json")))
    args.update(device=[DEV], compute_dtype="f32", batch_size=8, epochs=1, num_workers=0)
    args.update(over)
    return args


@pytest.fixture(scope="module")
def net12():
    from lpi_amd.retrieval.models.slinet import SliNet
    net = SliNet(vitb16_args()).to(DEV)
    for t in range(len(net.prompts)):
        for k, v in synth.prompt_factors(9, 16, CFG.vision_width, CFG.transformer_width, task=t).items():
            getattr(net.prompts[t], k).data = torch.from_numpy(v.copy()).to(DEV)
    net.numtask = 12
    net.train()
    for n, p in net.named_parameters():
        p.requires_grad_("prompts.11." in n)
    yield net
    del net
    torch.cuda.empty_cache()


def check_against(g, img_f, txt_f, losses, net):
    assert np.abs(img_f.detach().cpu().numpy() - g["img_f"]).max() < 1e-5
    assert np.abs(txt_f.detach().cpu().numpy() - g["txt_f"]).max() < 1e-5
    assert set(losses) == {"base_loss", "alignment_loss", "task_loss"}
    for k, v in losses.items():
        got = sum(float(p) for p in v) if isinstance(v, tuple) else float(v)
        assert abs(got - float(g[k])) < 1e-4 * max(1.0, abs(float(g[k]))), (k, got, float(g[k]))
    for k in synth.PROMPT_NAMES:
        assert rel_err(getattr(net.prompts[11], k).grad.cpu().numpy(), g["grad." + k]) <= 1e-3, k


def test_task_twelve_step_in_the_reference_order_matches_reference(golden, net12):
    g = golden("vitb16_task12")
    for p in net12.prompts[11].parameters():
        p.grad = None
    img = torch.from_numpy(synth.images(8, 224)).to(DEV)
    img_f, txt_f, vp, tp = net12(img, torch.from_numpy(g["token_ids"]))
    out = net12.cal_loss(img_f, txt_f, vp, tp)
    sum(v for v in out["loss"].values()).backward()
    torch.cuda.synchronize()
    check_against(g, img_f, txt_f, out["loss"], net12)


def test_task_twelve_fused_step_matches_reference_and_a_restored_checkpoint_drops_the_cached_rows(golden, net12):
    g = golden("vitb16_task12")
    img = torch.from_numpy(synth.images(8, 224)).to(DEV)
    ids = torch.from_numpy(g["token_ids"])
    for p in net12.prompts[11].parameters():
        p.grad = None
    out = net12.train_step(img, ids)
    torch.cuda.synchronize()
    check_against(g, out["image_features"], out["text_features"], out["loss"], net12)
    assert net12._task_term is not None and net12._task_term[0] == 12
    # the fused term caches the finished tasks' stacks; a restored checkpoint (in-place copy of the factors) must rebuild them
    saved = net12.trainable_state_dict()
    moved = {k: (v + 0.25 if k.startswith("prompts.3.") else v) for k, v in saved.items()}       # an EARLIER task's factors change
    net12.load_trainable_state_dict(moved)
    assert net12._task_term is None
    o2 = net12.train_step(img, ids)
    t_moved = sum(float(p) for p in o2["loss"]["task_loss"])
    net12.load_trainable_state_dict(saved)
    assert net12._task_term is None
    o3 = net12.train_step(img, ids)
    t_back = sum(float(p) for p in o3["loss"]["task_loss"])
    assert abs(t_back - float(g["task_loss"])) < 1e-4 and abs(t_moved - t_back) > 1e-4, (t_moved, t_back)
    # the non-fused path recomputes every step: the two must agree on the moved state as well
    net12.load_trainable_state_dict(moved)
    with torch.no_grad():
        ref_moved = 0.1 * float(net12.cal_task_loss(11, None, None))
    net12.load_trainable_state_dict(saved)
    assert abs(t_moved - ref_moved) < 1e-5 * max(1.0, abs(ref_moved)), (t_moved, ref_moved)
    # masters stay on the host, the trainable state is on the device
    assert net12.clip_model.visual.conv1.weight.device.type == "cpu" and net12.prompts[0].dim_1_share.device.type == "cuda"

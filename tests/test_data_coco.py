"""The COCO adapter of the plugin surface (lpi_amd/retrieval/utils/data.py: Coco / CocoEval, the reference's utils/data.py:160-382) on a
tiny COCO-format data set written on the fly: annotation filtering by task, the item contract of the training / evaluation loops, the
lookup tables itm_eval reads, caption normalisation, and the PIL restatement of the torchvision pipelines.  Host-side I/O: CPU only."""
import json
import os

import numpy as np
import pytest
import torch

PIL = pytest.importorskip("PIL")
from PIL import Image  # noqa: E402

from lpi_amd.retrieval.utils import data as D  # noqa: E402


@pytest.fixture(scope="module")
def coco(tmp_path_factory):
    root = tmp_path_factory.mktemp("coco")
    rng = np.random.default_rng(0)
    cats = [11, 11, 6, 3, 6, 1]                               # tasks 0, 0, 1, 2, 1, 11
    train, val = [], []
    for i, c in enumerate(cats):
        w, h = [(320, 240), (240, 320), (500, 375), (64, 48), (224, 224), (300, 600)][i]
        Image.fromarray(rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)).save(root / f"im{i}.png")
        train.append({"image": f"im{i}.png", "caption": f"A Photo, of  thing-{i}/x <person>!", "category": c, "image_id": f"coco_{i}"})
        val.append({"image": f"im{i}.png", "caption": [f"First caption {i}.", f"second: caption {i}"], "category": c, "image_id": i})
    (root / "train.json").write_text(json.dumps(train))
    (root / "val.json").write_text(json.dumps(val))
    return root


def test_pre_caption_and_task_order():
    assert D.pre_caption("A Photo, of  thing-1/x <person>!", 30) == "a photo of thing 1 x person"
    assert D.pre_caption("one two three four", 2) == "one two"
    with pytest.raises(ValueError):
        D.pre_caption("?!", 5)
    assert D.TASK_CATEGORIES == (11, 6, 3, 10, 5, 12, 7, 9, 2, 8, 4, 1)          # utils/data.py:233-249
    assert [D.task_of_category(c) for c in (11, 6, 3, 1)] == [0, 1, 2, 11] and D.task_of_category(99) == 0


def test_coco_training_items(coco):
    ds = D.Coco(image_root=str(coco), ann_file=str(coco / "train.json"), tasks=[0], prompt="X X ")
    assert len(ds) == 2 and len(ds.img_ids) == 2
    torch.manual_seed(0)
    img, cap, zero, task = ds[1]
    assert img.shape == (3, 224, 224) and img.dtype == torch.float32 and zero == 0 and task == 0
    assert cap == "X X a photo of thing 1 x person"
    lo = (0.0 - np.array(D.IMAGENET_MEAN)) / np.array(D.IMAGENET_STD)
    hi = (1.0 - np.array(D.IMAGENET_MEAN)) / np.array(D.IMAGENET_STD)
    for c in range(3):
        assert float(img[c].min()) >= lo[c] - 1e-5 and float(img[c].max()) <= hi[c] + 1e-5
    torch.manual_seed(0)
    again = ds[1][0]
    assert torch.equal(img, again)                                   # the augmentation draws from torch's RNG
    assert not torch.equal(img, ds[1][0])
    both = D.Coco(image_root=str(coco), ann_file=str(coco / "train.json"), tasks=[1, 11], replay_list=[ds.annotation[0]])
    assert [a["category"] for a in both.annotation] == [6, 6, 1, 11]
    assert both[2][3] == 11
    loader = torch.utils.data.DataLoader(both, batch_size=4, shuffle=False)
    images, captions, _, tasks = next(iter(loader))
    assert images.shape == (4, 3, 224, 224) and len(captions) == 4 and tasks.tolist() == [1, 1, 11, 0]


def test_coco_eval_tables_and_items(coco):
    ds = D.CocoEval(image_root=str(coco), ann_file=str(coco / "val.json"), tasks=np.arange(0, 3))
    assert len(ds) == 5 and ds.image == ["im0.png", "im1.png", "im2.png", "im3.png", "im4.png"]
    assert len(ds.text) == 10 and ds.text[0] == "first caption 0" and ds.text[1] == "second caption 0"
    assert ds.text_cat == [0, 0, 0, 0, 1, 1, 2, 2, 1, 1]
    assert ds.img2txt[2] == [4, 5] and ds.txt2img[5] == 2 and all(ds.txt2img[t] == i for i, ts in ds.img2txt.items() for t in ts)
    img, idx, task = ds[3]
    assert img.shape == (3, 224, 224) and idx == 3 and task == 2
    assert torch.equal(img, ds[3][0])                                # evaluation is deterministic
    ref_default = D.CocoEval(image_root=str(coco), ann_file=str(coco / "val.json"), tasks=[0], eval_transform='reference')
    assert ref_default.transform is D.train_transform and ds.transform is D.test_transform and ds.eval_transform == 'center'
    with pytest.raises(ValueError):
        D.CocoEval(image_root=str(coco), ann_file=str(coco / "val.json"), tasks=[0], eval_transform='random')
    rand = D.CocoEval(transform=D.train_transform, image_root=str(coco), ann_file=str(coco / "val.json"), tasks=[0])
    assert rand[0][0].shape == (3, 224, 224)


def test_transforms_match_a_direct_restatement(coco):
    """Resize(256) on the shorter side + CenterCrop(224): against PIL calls spelled out, incl. the rounding of the crop origin; the
    random crop stays inside the image and keeps the scale / ratio bounds."""
    im = Image.open(coco / "im2.png").convert("RGB")               # 500 x 375
    t = D.test_transform(im)
    r = im.resize((int(256 * 500 / 375), 256), Image.BILINEAR)
    left, top = int(round((r.size[0] - 224) / 2.0)), 16
    ref = np.asarray(r.crop((left, top, left + 224, top + 224)), dtype=np.float32) / 255.0
    ref = (ref - np.array(D.IMAGENET_MEAN, dtype=np.float32)) / np.array(D.IMAGENET_STD, dtype=np.float32)
    assert np.abs(t.permute(1, 2, 0).numpy() - ref).max() < 1e-6
    small = Image.open(coco / "im3.png").convert("RGB")             # 64 x 48: upsampled
    assert D.test_transform(small).shape == (3, 224, 224)
    torch.manual_seed(3)
    for _ in range(20):
        assert D.train_transform(im).shape == (3, 224, 224)


def test_sprompts_selects_the_coco_datasets(coco, monkeypatch):
    from lpi_amd.retrieval.methods import sprompt as S
    m = S.SPrompts.__new__(S.SPrompts)
    m.args = {"image_root": str(coco), "annotation_train_root": str(coco / "train.json"), "annotation_val_root": str(coco / "val.json")}
    tr, te = S.SPrompts._datasets(m, 1)
    assert isinstance(tr, D.Coco) and isinstance(te, D.CocoEval) and len(tr) == 2 and len(te) == 4
    m.args["dataset_impl"] = "nope"
    with pytest.raises(ValueError):
        S.SPrompts._datasets(m, 0)


def test_uint8_pixel_format_is_the_f32_pipeline_before_totensor_and_normalize(tmp_path):
    """pixel_format='u8' (round 5): the datasets hand over the decoded CHW bytes and the GPU applies ToTensor + Normalize through a 3 x 256 table
    (lpi_patchify_u8).  On the host: the table equals the loader's own arithmetic for every byte and channel, bit for bit, and a u8 item normalised with
    it IS the f32 item."""
    import numpy as np
    import torch
    from lpi_amd.engine import make_pixel_lut
    from lpi_amd.retrieval.utils import data as D
    lut = make_pixel_lut()
    for c in range(3):
        ramp = torch.arange(256, dtype=torch.uint8).view(1, 16, 16).expand(3, 16, 16).contiguous()
        ref = D.normalise_u8(ramp)[c].reshape(-1)
        assert torch.equal(lut[c], ref)
    try:
        from PIL import Image
    except ImportError:
        return
    rng = np.random.default_rng(0)
    img = Image.fromarray(rng.integers(0, 256, (300, 260, 3), dtype=np.uint8))
    u8 = D.test_transform(img, pixel_format="u8")
    f32 = D.test_transform(img)
    assert u8.dtype == torch.uint8 and u8.shape == (3, 224, 224) and f32.dtype == torch.float32
    assert torch.equal(D.normalise_u8(u8), f32)
    gathered = torch.stack([lut[c][u8[c].long()] for c in range(3)])          # what the kernel computes
    assert torch.equal(gathered, f32)
    ds = D.SyntheticCoco(5, [0], 32, pixel_format="u8", image_pool=3)
    assert ds[4][0].dtype == torch.uint8 and ds[4][0].data_ptr() == ds[1][0].data_ptr()

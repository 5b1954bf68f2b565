"""Dataset I/O contract of the retrieval path (utils/data.py:160-382 of the reference): train items are
``(image[3,224,224] f32 ImageNet-normalised, caption, 0, task)``, eval items ``(image, img_id, task)`` with the lookup tables
``text, image, text_cat, txt2img, img2txt`` on the dataset object.  COCO itself is not available offline, so the default
implementation is synthetic (captions are delivered as ready token ids).  ``Coco`` / ``CocoEval`` read the reference's annotation
format (a JSON list of ``{"image", "caption", "category", "image_id"}`` records, ``caption`` a string for training and a list for
evaluation) from ``image_root`` / ``ann_file`` with PIL — the torchvision pipelines of the reference (RandomResizedCrop +
RandomHorizontalFlip / Resize(256) + CenterCrop(224), ToTensor, ImageNet Normalize) are restated here on PIL + torch, because
torchvision is not part of this environment.  Host-side I/O only: nothing here is on the timed path."""
import json
import math
import os
import re

import numpy as np
import torch
from torch.utils.data import Dataset

from lpi_amd import synth


def pre_caption(caption, max_words):
    """utils/data.py:160-185."""
    caption = re.sub(r"([,.'!?\"()*#:;~])", '', caption.lower()).replace('-', ' ').replace('/', ' ').replace('<person>', 'person')
    caption = re.sub(r"\s{2,}", ' ', caption).rstrip('\n').strip(' ')
    words = caption.split(' ')
    if len(words) > max_words:
        caption = ' '.join(words[:max_words])
    if not len(caption):
        raise ValueError("pre_caption yields invalid text")
    return caption


class SyntheticCoco(Dataset):
    """Training pairs of one task: N(0,1) images and random captions (SURVEY.md section 8(d) recipe).

    captions = 'ids' (default): a caption is its row of ready token ids [77] (no tokenizer, no merge table needed); 'strings': a caption is a STRING of
    COCO-like words (lpi_amd.synth_bpe.captions) — the item then has exactly the reference's structure (utils/data.py:376-382: f32 image, str, 0, task) and
    the step tokenises it like PromptLearner.forward does.
    pixel_format = 'u8': items carry uint8 CHW pixels (uniform bytes) and ToTensor + Normalize run on the GPU (lpi_patchify_u8).
    image_pool = K > 0: the K distinct images are generated once and item i returns pool image i % K (a VIEW: the collate / pipeline copies it) — an
    item costs nothing, so that a throughput measurement of the training loop times the loop and not numpy's generator (150 k normals per image)."""

    def __init__(self, n, tasks, resolution=224, seed=0, captions="ids", image_pool=0, pixel_format="f32"):
        if captions not in ("ids", "strings"):
            raise ValueError(f"captions must be 'ids' or 'strings', not {captions!r}")
        if pixel_format not in ("f32", "u8"):
            raise ValueError(f"pixel_format must be 'f32' or 'u8', not {pixel_format!r}")
        self.pixel_format = pixel_format
        self.n, self.tasks, self.res = n, list(tasks), resolution
        self.seed = seed
        if captions == "ids":
            self.ids = torch.from_numpy(synth.token_ids(n, seed=synth.TOKEN_SEED + 17 * seed))
            self.captions = None
        else:
            from lpi_amd.synth_bpe import captions as make
            self.ids, self.captions = None, make(n, seed=synth.TOKEN_SEED + 17 * seed)
        k = min(int(image_pool), n) if image_pool else 0
        if pixel_format == "u8":          # uniform bytes: "decoded pixels"; pixel_format='f32' of the same dataset = their ToTensor + Normalize
            k = k or n
            self.pool = torch.from_numpy(synth._rng(synth.IMAGE_SEED + seed, f"u8pool{k}").integers(0, 256, (k, 3, resolution, resolution), dtype=np.uint8))
        else:
            self.pool = None if k <= 0 else torch.from_numpy(synth.normal(synth.IMAGE_SEED + seed, f"pool{k}", (k, 3, resolution, resolution)))

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        if self.pool is not None:
            img = self.pool[i % self.pool.shape[0]]
        else:
            img = torch.from_numpy(synth.normal(synth.IMAGE_SEED + self.seed, f"img{i}", (3, self.res, self.res)))
        return img, (self.ids[i] if self.captions is None else self.captions[i]), 0, self.tasks[0]


def collate_keep_images(batch):
    """default_collate for everything but the images, which stay a LIST of [3,R,R] tensors: lpi_amd.pipeline gathers them straight into its pinned staging
    buffer (one copy, several threads) instead of torch.stack into pageable memory followed by a second copy (single-process loaders only: tensors that
    cross a worker boundary must be stacked there)."""
    from torch.utils.data import default_collate
    cols = list(zip(*batch))
    return [list(cols[0])] + [default_collate(list(c)) for c in cols[1:]]


class SyntheticCocoEval(Dataset):
    """Eval set over tasks 0..t: n_img images per task, `cpi` captions per image."""

    def __init__(self, n_img_per_task, tasks, cpi=2, resolution=224, seed=1):
        self.res, self.seed = resolution, seed
        self.image, self.text, self.text_cat, self.img_cat = [], [], [], []
        self.txt2img, self.img2txt = {}, {}
        for t in tasks:
            for _ in range(n_img_per_task):
                i = len(self.image)
                self.image.append(i)
                self.img_cat.append(int(t))
                self.img2txt[i] = []
                for _ in range(cpi):
                    j = len(self.text)
                    self.text.append(j)
                    self.text_cat.append(int(t))
                    self.txt2img[j] = i
                    self.img2txt[i].append(j)
        self.ids = torch.from_numpy(synth.token_ids(len(self.text), seed=synth.TOKEN_SEED + 1000 + seed))
        self.text = self.ids          # "texts" are token-id rows; slicing works like the reference's list slicing

    def __len__(self):
        return len(self.image)

    def __getitem__(self, i):
        img = torch.from_numpy(synth.normal(synth.IMAGE_SEED + 500 + self.seed, f"img{i}", (3, self.res, self.res)))
        return img, i, self.img_cat[i]


# ---------------------------------------------------------------------------------------------- real COCO (utils/data.py:186-382)
# task t of the 12-task protocol holds COCO super-category TASK_CATEGORIES[t] (utils/data.py:233-249, 338-353)
TASK_CATEGORIES = (11, 6, 3, 10, 5, 12, 7, 9, 2, 8, 4, 1)
IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def task_of_category(category: int) -> int:
    """The reference's loop `for z in range(len(tasks)): if category in tasks[z]: new_category = z` (0 when absent)."""
    return TASK_CATEGORIES.index(category) if category in TASK_CATEGORIES else 0


def _pil():
    try:
        from PIL import Image
    except ImportError as e:          # loud: there is no silent fallback to synthetic data
        raise ImportError("the COCO datasets need Pillow (PIL) to decode images; use dataset_impl='synthetic' without it") from e
    return Image


def _to_u8_chw(img):
    """The decoded pixels as the GPU takes them (pixel_format='u8'): HWC uint8 -> CHW uint8; ToTensor + Normalize then run inside lpi_patchify_u8."""
    return torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).contiguous()


def normalise_u8(u8):
    """ToTensor + Normalize on a CHW (or BCHW) uint8 tensor, in the operations of _to_normalised_tensor — the f32 image the 'f32' pixel format delivers."""
    a = u8.float().div_(255.0)
    shape = (3, 1, 1)
    return (a - torch.tensor(IMAGENET_MEAN).view(shape)) / torch.tensor(IMAGENET_STD).view(shape)


def _to_normalised_tensor(img):
    """ToTensor + Normalize(ImageNet) (utils/data.py:201-204): HWC uint8 -> CHW f32 in [0,1], then (x - mean) / std."""
    a = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float().div_(255.0)
    mean = torch.tensor(IMAGENET_MEAN).view(3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(3, 1, 1)
    return (a - mean) / std


def train_transform(img, size=224, scale=(0.08, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), pixel_format="f32"):
    """RandomResizedCrop(size) + RandomHorizontalFlip + ToTensor + Normalize (utils/data.py:193-204), torch RNG: a crop of random area
    (scale x image area) and log-uniform aspect ratio, ten attempts, else the largest centred crop inside the ratio bounds; bilinear."""
    Image = _pil()
    w, h = img.size
    area = w * h
    box = None
    for _ in range(10):
        target = area * float(torch.empty(1).uniform_(scale[0], scale[1]))
        logr = float(torch.empty(1).uniform_(math.log(ratio[0]), math.log(ratio[1])))
        ar = math.exp(logr)
        cw, ch = int(round(math.sqrt(target * ar))), int(round(math.sqrt(target / ar)))
        if 0 < cw <= w and 0 < ch <= h:
            top = int(torch.randint(0, h - ch + 1, (1,)))
            left = int(torch.randint(0, w - cw + 1, (1,)))
            box = (left, top, left + cw, top + ch)
            break
    if box is None:
        in_ratio = w / h
        if in_ratio < ratio[0]:
            cw, ch = w, int(round(w / ratio[0]))
        elif in_ratio > ratio[1]:
            cw, ch = int(round(h * ratio[1])), h
        else:
            cw, ch = w, h
        left, top = (w - cw) // 2, (h - ch) // 2
        box = (left, top, left + cw, top + ch)
    img = img.crop(box).resize((size, size), Image.BILINEAR)
    if float(torch.rand(1)) < 0.5:
        img = img.transpose(Image.FLIP_LEFT_RIGHT)
    return _to_u8_chw(img) if pixel_format == "u8" else _to_normalised_tensor(img)


def test_transform(img, resize=256, size=224, pixel_format="f32"):
    """Resize(256) (shorter side, bilinear) + CenterCrop(224) + ToTensor + Normalize (utils/data.py:197-204)."""
    Image = _pil()
    w, h = img.size
    if w <= h:
        nw, nh = resize, int(resize * h / w)
    else:
        nw, nh = int(resize * w / h), resize
    img = img.resize((nw, nh), Image.BILINEAR)
    left, top = int(round((nw - size) / 2.0)), int(round((nh - size) / 2.0))
    img = img.crop((left, top, left + size, top + size))
    return _to_u8_chw(img) if pixel_format == "u8" else _to_normalised_tensor(img)


def _load(image_root, name, transform):
    Image = _pil()
    with Image.open(os.path.join(image_root, name)) as im:
        return transform(im.convert("RGB"))


class Coco(Dataset):
    """Training pairs of the given tasks (utils/data.py:308-382): item = (image, prompt + pre_caption(caption), 0, task)."""

    def __init__(self, transform=None, image_root=None, ann_file=None, max_words=30, prompt='', tasks=(0,), replay_list=(), pixel_format="f32"):
        _pil()
        if pixel_format not in ("f32", "u8"):
            raise ValueError(f"pixel_format must be 'f32' or 'u8', not {pixel_format!r}")
        if transform is None and pixel_format == "u8":
            transform = lambda im: train_transform(im, pixel_format="u8")  # noqa: E731
        with open(ann_file, 'r') as f:
            records = json.load(f)
        cats = {TASK_CATEGORIES[int(t)] for t in tasks}
        self.transform = transform or train_transform
        self.image_root, self.max_words, self.prompt = image_root, max_words, prompt
        self.img_ids = {}
        self.annotation = []
        for ann in records:
            if ann['category'] in cats:
                self.img_ids.setdefault(ann['image_id'], len(self.img_ids))
                self.annotation.append(ann)
        self.annotation += list(replay_list)

    def __len__(self):
        return len(self.annotation)

    def __getitem__(self, index):
        ann = self.annotation[index]
        image = _load(self.image_root, ann['image'], self.transform)
        return image, self.prompt + pre_caption(ann['caption'], self.max_words), 0, task_of_category(ann['category'])


class CocoEval(Dataset):
    """Evaluation images of tasks 0..t with every caption of every image in the lookup tables the scoring loop reads
    (utils/data.py:186-306; sprompt.py:433-548): text, text_cat, image, txt2img, img2txt; item = (image, image index, task)."""

    def __init__(self, transform=None, image_root=None, ann_file=None, max_words=30, tasks=(0,), eval_transform='center', pixel_format="f32"):
        _pil()
        if pixel_format not in ("f32", "u8"):
            raise ValueError(f"pixel_format must be 'f32' or 'u8', not {pixel_format!r}")
        if transform is None and pixel_format == "u8":
            base = test_transform if eval_transform == 'center' else train_transform
            transform = lambda im: base(im, pixel_format="u8")  # noqa: E731
        with open(ann_file, 'r') as f:
            records = json.load(f)
        cats = {TASK_CATEGORIES[int(t)] for t in tasks}
        # eval_transform (args['eval_transform'], recorded in the result file): 'center' = the deterministic Resize + CenterCrop the reference
        # defines for evaluation; 'reference' = what the reference's CocoEval actually applies when constructed as sprompt.py:169 does — its default
        # argument is the TRAINING transform (utils/data.py:206: RandomResizedCrop + flip), so R@K of a parity run against the reference code as
        # written needs 'reference'.  An explicit `transform` wins over both.
        if eval_transform not in ('center', 'reference'):
            raise ValueError(f"eval_transform must be 'center' or 'reference', not {eval_transform!r}")
        self.eval_transform = eval_transform
        self.transform = transform or (test_transform if eval_transform == 'center' else train_transform)
        self.image_root, self.max_words = image_root, max_words
        self.ann = [a for a in records if a['category'] in cats]
        self.text, self.text_cat, self.image = [], [], []
        self.txt2img, self.img2txt = {}, {}
        for img_id, ann in enumerate(self.ann):
            self.image.append(ann['image'])
            self.img2txt[img_id] = []
            task = task_of_category(ann['category'])
            for caption in ann['caption']:
                txt_id = len(self.text)
                self.text.append(pre_caption(caption, self.max_words))
                self.text_cat.append(task)
                self.img2txt[img_id].append(txt_id)
                self.txt2img[txt_id] = img_id

    def __len__(self):
        return len(self.ann)

    def __getitem__(self, index):
        ann = self.ann[index]
        return _load(self.image_root, ann['image'], self.transform), index, task_of_category(ann['category'])

"""Dataset I/O contract of the retrieval path (utils/data.py:160-382 of the reference): train items are
``(image[3,224,224] f32 ImageNet-normalised, caption, 0, task)``, eval items ``(image, img_id, task)`` with the lookup tables
``text, image, text_cat, txt2img, img2txt`` on the dataset object.  COCO itself is not available offline, so the default
implementation is synthetic (captions are delivered as ready token ids); ``pre_caption`` is the reference's caption
normaliser for callers that bring real annotations."""
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
    """Training pairs of one task: N(0,1) images and random caption token ids (SURVEY.md section 8(d) recipe)."""

    def __init__(self, n, tasks, resolution=224, seed=0):
        self.n, self.tasks, self.res = n, list(tasks), resolution
        self.ids = torch.from_numpy(synth.token_ids(n, seed=synth.TOKEN_SEED + 17 * seed))
        self.seed = seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        img = torch.from_numpy(synth.normal(synth.IMAGE_SEED + self.seed, f"img{i}", (3, self.res, self.res)))
        return img, self.ids[i], 0, self.tasks[0]


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

"""SliNet — the network plugin surface of the reference (models/slinet.py:12-234), backed by the MI355X HIP engine.

Same constructor argument (the flat ``args`` dict of main.py), same attributes and methods the learner uses:
``forward(image, text) -> (img_f, txt_f, vis_prompt, txt_prompt)``, ``cal_loss(...) -> {'loss': {...}}``,
``extract_vector``, ``extract_textual_vector``, ``visual_interface``, ``textual_interface``, ``update_fc``, ``copy``,
``freeze``, ``numtask``, ``class_num``, ``feature_dim`` and the parameter naming contract
``prompts.{task}.dim_1_share|dim_2_visual|dim_2_textual|dim_3_visual|dim_3_textual`` (substring filter sprompt.py:235).

Differences that are deliberate:
  * only ``prompt_type == 'lpi'`` / ``net_type == 'slip'`` is built (the hot path); others raise ValueError;
  * weights: ``args['clip_state_dict']`` (a CLIP state dict or a path to one) or, without network access, the deterministic
    synthetic weights of ``lpi_amd.synth`` — the reference downloads them (prompt_learner.py:10-13);
  * ``text`` may be a list of captions (tokenised on the host like PromptLearner.forward) or a LongTensor [B,77] of ids;
  * extra keys: ``compute_dtype`` ('f32' parity mode | 'bf16' throughput mode), ``honor_prompt_depth`` (default False: the
    shipped reference never reads ``prompt_depth`` and behaves as depth 1 — SURVEY.md F1), ``r`` (default 4).
"""
import copy
import os

import numpy as np
import torch
import torch.nn as nn

from lpi_amd import synth
from lpi_amd.functional import AlignLossFn, ClipLossFn, EncodeBothFn
from lpi_amd.retrieval.loss.loss import ClipLoss, nt_bxent_loss
from lpi_amd.retrieval.models.clip.prompt_learner import PromptLearner, cfgc
from lpi_amd.retrieval.models.prompts.prompts import DecomposedPrompt

_HERE = os.path.dirname(os.path.abspath(__file__))


def _load_state_dict(args, cfg):
    sd = args.get("clip_state_dict")
    if isinstance(sd, str):
        obj = torch.load(sd, map_location="cpu")
        sd = obj.state_dict() if hasattr(obj, "state_dict") else obj
    if sd is None:
        sd = synth.clip_state_dict(cfg)
    return sd


class SliNet(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.cfg = cfgc()
        self.args = args
        self.cfg.backbonename = args["backbonename"]
        self.cfg.NCTX = args["NCTX"]
        self.cfg.CTXINIT = args["CTXINIT"]
        self.cfg.CSC = args["CSC"]
        self.cfg.CLASS_TOKEN_POSITION = args["CLASS_TOKEN_POSITION"]
        if args["prompt_type"] != "lpi":
            raise ValueError("lpi_amd builds prompt_type 'lpi' only (the hot path); got {}".format(args["prompt_type"]))
        if args["backbonename"] not in synth.CONFIGS:
            raise KeyError(args["backbonename"])
        self.clip_cfg = synth.CONFIGS[args["backbonename"]]
        if args["visual_dim"] != self.clip_cfg.vision_width or args["textual_dim"] != self.clip_cfg.transformer_width:
            raise ValueError("visual_dim / textual_dim do not match the backbone")
        self.compute_dtype = args.get("compute_dtype", "f32")
        self.depth = int(args["prompt_depth"]) if args.get("honor_prompt_depth", False) else 1
        self.prompts = nn.ModuleList([
            DecomposedPrompt(9, args["prompt_length"], args["visual_dim"], args["textual_dim"], r=args.get("r", 4))   # slinet.py:44-47
            for _ in range(args["total_sessions"])
        ])
        self.classifier_pool = [PromptLearner(self.cfg, self.clip_cfg.context_length) for _ in range(args["total_sessions"])]
        self.class_num = 2
        self.numtask = 0
        self.loss = ClipLoss()
        self.alignment_loss = ClipLoss()
        self.all_keys = []
        self.engine = None
        self.exchange = None        # set to a lpi_amd.dp.Exchange for data-parallel training
        self.dtype = torch.float32

    # ------------------------------------------------------------------ device / engine
    def _ensure_engine(self, device=None):
        if self.engine is None:
            from lpi_amd.engine import DualEncoder
            dev = torch.device(device) if device is not None else next(self.parameters()).device
            self.engine = DualEncoder(self.clip_cfg, _load_state_dict(self.args, self.clip_cfg), dtype=self.compute_dtype,
                                      device=dev, n_ctx=self.cfg.NCTX)
            self.logit_scale = self.engine.logit_scale
        return self.engine

    def to(self, *a, **k):
        out = super().to(*a, **k)
        dev = next(self.parameters()).device
        if dev.type == "cuda":
            self._ensure_engine(dev)
        return out

    @property
    def feature_dim(self):
        return self.clip_cfg.embed_dim

    def _ids(self, text, pool_idx):
        ids = self.classifier_pool[pool_idx](text)
        if not ids.is_cuda and self.args.get("trim_text", True):
            # rows behind a caption's EOT are dead under the causal mask: exact, and free here because the tokenizer ran on the host.
            # pack_text (default): every caption cut at its OWN end, the batch packed (engine.PackedIds); else cut at the longest one
            from lpi_amd.engine import PackedIds, trim_token_ids
            if self.args.get("pack_text", True):
                return PackedIds(ids).to(self.engine.device)
            ids = trim_token_ids(ids).contiguous()
        return ids.to(self.engine.device)

    # ------------------------------------------------------------------ slinet.py:85-107
    def extract_vector(self, image):
        return self._ensure_engine().encode_image(image, None)

    extract_visual_vector = extract_vector

    def extract_textual_vector(self, text):
        eng = self._ensure_engine()
        return eng.encode_text(self._ids(text, self.numtask - 1), None)

    # ------------------------------------------------------------------ slinet.py:109-135
    def forward(self, image, text):
        eng = self._ensure_engine()
        visual_prompt, textual_prompt = self.prompts[self.numtask - 1]()
        bs = image.shape[0]
        ids = self._ids(text, self.numtask - 1)
        # the two towers in lock step (one autograd node): their GEMMs of the same layer op go out as one grouped launch
        image_features, text_features = EncodeBothFn.apply(eng, image, ids, visual_prompt, textual_prompt, self.depth)
        return image_features, text_features, visual_prompt.expand(bs, -1, -1, -1), textual_prompt.expand(bs, -1, -1, -1)

    # ------------------------------------------------------------------ slinet.py:137-183
    @staticmethod
    def _dense(p):
        if p.dim() == 4:
            return p[0] if p.stride(0) == 0 else p.mean(0)     # mean over a stride-0 batch == the dense tensor
        return p

    def cal_loss(self, image_featuers, text_features, visual_prompt, textual_prompt):
        eng = self._ensure_engine()
        gather = self.exchange.gather if self.exchange is not None else None
        losses = {"base_loss": ClipLossFn.apply(image_featuers, text_features, eng.logit_scale_exp, gather, self.exchange)}
        vis, txt = self._dense(visual_prompt), self._dense(textual_prompt)
        losses["alignment_loss"] = AlignLossFn.apply(vis, txt, 0.01, 0.1)
        if self.numtask != 1:
            losses["task_loss"] = 0.1 * self.cal_task_loss(self.numtask - 1, None, None)
        return {"loss": losses}

    def cal_task_loss(self, task_id, visual_prompt, textual_prompt):
        path = "./MID/task_sim_matrix.txt"                          # cwd-relative like slinet.py:171
        if not os.path.exists(path):
            path = os.path.join(os.path.dirname(_HERE), "MID", "task_sim_matrix.txt")
        sim = torch.tensor(np.loadtxt(path)[:task_id + 1, :task_id + 1])
        dev = self.prompts[0].dim_1_share.device
        target = (sim > 0.4).type(torch.int).to(dev)
        dense = [self.prompts[i]() for i in range(task_id + 1)]
        vs = torch.stack([v.reshape(-1) for v, _ in dense])
        ts = torch.stack([t.reshape(-1) for _, t in dense])
        # HIP kernels (lpi_nt_bxent_fwd_bwd); raises without a GPU like every other op of the path
        return (nt_bxent_loss(vs, target, 0.001, task_id) + nt_bxent_loss(ts, target, 0.001, task_id)) / 2

    # ------------------------------------------------------------------ slinet.py:185-220
    def textual_interface(self, text, text_category):
        eng = self._ensure_engine()
        if self.training:
            return eng.encode_text(self._ids(text, self.numtask - 1), None)
        with torch.no_grad():
            stack = torch.stack([p()[1] for p in self.prompts], 0)[text_category.to(stack_device(self))]
        return eng.encode_text(self._ids(text, 0), stack, self.depth)

    def visual_interface(self, image, image_category):
        eng = self._ensure_engine()
        with torch.no_grad():
            stack = torch.stack([p()[0] for p in self.prompts], 0)[image_category.to(stack_device(self))]
        return eng.encode_image(image, stack, self.depth)

    # ------------------------------------------------------------------ slinet.py:223-234
    def update_fc(self, nb_classes):
        self.numtask += 1

    def copy(self):
        """Deep copy of the trainable state; the frozen engine (weights + workspace arena) is shared, not duplicated."""
        eng, self.engine = self.engine, None
        try:
            new = copy.deepcopy(self)
        finally:
            self.engine = eng
        new.engine = eng
        return new

    def freeze(self):
        for param in self.parameters():
            param.requires_grad = False
        self.eval()
        return self


def stack_device(net):
    return net.prompts[0].dim_1_share.device

"""SliNet — the network plugin surface of the reference (models/slinet.py:12-234), backed by the MI355X HIP engine.

Same constructor argument (the flat ``args`` dict of main.py), same attributes and methods the learner uses:
``forward(image, text) -> (img_f, txt_f, vis_prompt, txt_prompt)``, ``cal_loss(...) -> {'loss': {...}}``,
``extract_vector``, ``extract_textual_vector``, ``visual_interface``, ``textual_interface``, ``update_fc``, ``copy``,
``freeze``, ``numtask``, ``class_num``, ``feature_dim`` and the parameter naming contract
``prompts.{task}.dim_1_share|dim_2_visual|dim_2_textual|dim_3_visual|dim_3_textual`` (substring filter sprompt.py:235).

Differences that are deliberate:
  * only ``prompt_type == 'lpi'`` / ``net_type == 'slip'`` is built (the hot path); others raise ValueError;
  * weights: ``args['clip_state_dict']`` — a CLIP state dict, or the path of OpenAI's file as the reference downloads it (a TorchScript archive, or a
    torch.save'd state dict; fp16 or fp32; prompt_learner.py:10-40) — with the architecture inferred from the tensor shapes exactly as build_model does
    (model.py:418-445; lpi_amd/checkpoint.py); without it, the deterministic synthetic weights of ``lpi_amd.synth`` (there is no network here);
  * the frozen CLIP tensors are registered as (frozen) parameters under ``clip_model.*`` and the PromptLearners' unused ``ctx`` vectors under
    ``classifier_pool.N.ctx``, so that ``count_parameters`` (trainer.py:50-51) and ``state_dict()`` see the reference network's state (149.78 M);
  * ``train_step(image, text)``: forward + cal_loss + sum + backward of the hot loop (sprompt.py:303-310) as ONE fused call (lpi_amd.step.train_step: the
    loss kernels seed the towers' backward, no scalar loss graph) — what SPrompts.train_function runs; forward / cal_loss stay for every other caller;
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
from lpi_amd.checkpoint import ParamTree, infer_config, load_clip_state_dict, same_architecture
from lpi_amd.functional import AlignLossFn, ClipLossFn, EncodeBothFn
from lpi_amd.retrieval.loss.loss import ClipLoss, nt_bxent_loss
from lpi_amd.retrieval.models.clip.prompt_learner import PromptLearner, cfgc
from lpi_amd.retrieval.models.prompts.prompts import DecomposedPrompt

_HERE = os.path.dirname(os.path.abspath(__file__))


def _resolve_weights(args):
    """-> (ClipConfig, state dict).  A checkpoint decides the architecture (build_model takes every size from the tensors' shapes, model.py:419-441);
    ``backbonename`` must agree with it where this package knows the name.  Without a checkpoint: the synthetic weights of the named config."""
    name = args["backbonename"]
    src = args.get("clip_state_dict")
    if src is None:
        if name not in synth.CONFIGS:
            raise KeyError(name)
        cfg = synth.CONFIGS[name]
        return cfg, synth.clip_state_dict(cfg)
    sd = load_clip_state_dict(src, trusted=bool(args.get("clip_checkpoint_trusted", False)))
    cfg = infer_config(sd, name)
    known = synth.CONFIGS.get(name)
    if known is not None and not same_architecture(known, cfg):
        raise ValueError(f"clip_state_dict holds a {cfg.as_clip_args()} CLIP, backbonename {name!r} names {known.as_clip_args()}")
    return cfg, sd


class _TaskTerm:
    """The task loss of a continual session (slinet.py:160-162, 167-183) inside the fused step: 0.1 * (nt_bxent(V) + nt_bxent(T)) / 2 over the stacked,
    flattened prompt stacks of tasks 0..t.  Rows 0..t-1 (the finished tasks, frozen: sprompt.py:230-237) are reconstructed ONCE; row t is copied from the
    step's own stacks; the gradient w.r.t. row t is ADDED onto the towers' seeded prompt-gradient buffers (lpi_nt_bxent_fwd_bwd, accumulate = 1)."""

    def __init__(self, net, task_id):
        from lpi_amd import _lib
        self._call = _lib.call
        dev = net.prompts[0].dim_1_share.device
        self.t = task_id
        self.target = net._task_target(task_id).to(device=dev, dtype=torch.int32).contiguous()
        with torch.no_grad():
            dense = [net.prompts[i]() for i in range(task_id + 1)]
            self.Xv = torch.stack([v.reshape(-1) for v, _ in dense]).contiguous()
            self.Xt = torch.stack([t.reshape(-1) for _, t in dense]).contiguous()
        T = task_id + 1
        self.scratch = [torch.empty(2 * T * T, device=dev) for _ in range(2)]

    def __call__(self, vis, txt, dvis, dtxt, scale):
        s = torch.cuda.current_stream().cuda_stream
        T, t = self.t + 1, self.t
        out = []
        for X, cur, dbuf, scr in ((self.Xv, vis, dvis, self.scratch[0]), (self.Xt, txt, dtxt, self.scratch[1])):
            D = X.shape[1]
            self._call("lpi_copy_rows", 1, D, cur.contiguous(), D, X[t], D, s)
            loss = torch.empty(1, device=X.device)
            # weight: 0.1 (slinet.py:162) x 1/2 (the mean over the two modalities, :183) x the data-parallel 1/W; the VALUE is reported unscaled by W
            self._call("lpi_nt_bxent_fwd_bwd", T, D, t, X, self.target, 0.001, 0.05 * scale, loss, dbuf, 1, scr, s)
            out.append(loss if scale == 1.0 else loss / scale)
        return tuple(out)


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
        self.clip_cfg, sd = _resolve_weights(args)
        if args["visual_dim"] != self.clip_cfg.vision_width or args["textual_dim"] != self.clip_cfg.transformer_width:
            raise ValueError("visual_dim / textual_dim do not match the backbone")
        self.compute_dtype = args.get("compute_dtype", "f32")
        self.depth = int(args["prompt_depth"]) if args.get("honor_prompt_depth", False) else 1
        self.prompts = nn.ModuleList([
            DecomposedPrompt(9, args["prompt_length"], args["visual_dim"], args["textual_dim"], r=args.get("r", 4))   # slinet.py:44-47
            for _ in range(args["total_sessions"])
        ])
        self.classifier_pool = nn.ModuleList([PromptLearner(self.cfg, self.clip_cfg.context_length, ctx_dim=self.clip_cfg.transformer_width)
                                              for _ in range(args["total_sessions"])])
        # the frozen backbone as module state (the reference's self.clip_model, slinet.py:24-25): f32 masters; the engine keeps its own operand copies
        self.clip_model = ParamTree(sd)
        self.class_num = 2
        self.numtask = 0
        self.loss = ClipLoss()
        self.alignment_loss = ClipLoss()
        self.all_keys = []
        self.engine = None
        self.exchange = None        # set to a lpi_amd.dp.Exchange for data-parallel training
        self.dtype = torch.float32
        self._task_term = None      # (numtask, _TaskTerm) of the fused step

    # ------------------------------------------------------------------ device / engine
    def _ensure_engine(self, device=None):
        if self.engine is None:
            from lpi_amd.engine import DualEncoder
            dev = torch.device(device) if device is not None else self.prompts[0].dim_1_share.device
            from lpi_amd.engine import EngineOptions
            eo = self.args.get("engine_options")      # None, an EngineOptions, or a dict of its fields (the config file's form)
            if isinstance(eo, dict):
                eo = EngineOptions.from_env(**eo)
            self.engine = DualEncoder(self.clip_cfg, self.clip_model.state_dict(), dtype=self.compute_dtype, device=dev, n_ctx=self.cfg.NCTX, options=eo)
            self.logit_scale = self.engine.logit_scale
        return self.engine

    def _apply(self, fn, recurse=True):
        """.to() / .cuda() / .float() move the TRAINABLE state; the frozen f32 masters (clip_model: 0.6 GB for ViT-B/16, 1.7 GB for ViT-L/14) stay where the
        checkpoint put them — the host.  The engine holds its own operand copies on the device and never reads the masters after it is built, so a device
        copy of them would be HBM that nothing uses; state_dict() / named_parameters() / count_parameters still see all 149.78 M of them."""
        clip = self._modules.pop("clip_model", None)
        try:
            return super()._apply(fn, recurse)
        finally:
            if clip is not None:
                self._modules["clip_model"] = clip

    def to(self, *a, **k):
        out = super().to(*a, **k)
        dev = self.prompts[0].dim_1_share.device
        if dev.type == "cuda":
            self._ensure_engine(dev)
        return out

    @property
    def feature_dim(self):
        return self.clip_cfg.embed_dim

    def _shared_rows(self):
        """Rows of the text tower's SHARED PREFIX in the training forward (engine.PackedIds(shared=...)): SOT and the n_ctx context slots, which hold the
        same rows for every sample because the prompts are broadcast over the batch (slinet.py:119-130) — 0 with args['share_text_prefix'] = False.  Exact in
        every operand mode: the f32 parity mode meets the reference fixtures' 1e-4 bar on this layout (tests/test_shared_prefix_gpu.py)."""
        if not self.args.get("share_text_prefix", True):
            return 0
        return 1 + self.cfg.NCTX

    def _ids(self, text, pool_idx, shared=0):
        ids = self.classifier_pool[pool_idx](text)
        if not ids.is_cuda and self.args.get("trim_text", True):
            # rows behind a caption's EOT are dead under the causal mask: exact, and free here because the tokenizer ran on the host.
            # pack_text (default): every caption cut at its OWN end, the batch packed (engine.PackedIds); else cut at the longest one
            from lpi_amd.engine import PackedIds, trim_token_ids
            if self.args.get("pack_text", True):
                return PackedIds(ids, shared).to(self.engine.device)
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
        ids = self._ids(text, self.numtask - 1, self._shared_rows())
        # the two towers in lock step (one autograd node): their GEMMs of the same layer op go out as one grouped launch
        image_features, text_features = EncodeBothFn.apply(eng, image, ids, visual_prompt, textual_prompt, self.depth)
        return image_features, text_features, visual_prompt.expand(bs, -1, -1, -1), textual_prompt.expand(bs, -1, -1, -1)

    # ------------------------------------------------------------------ the hot loop's step, fused (sprompt.py:303-310)
    def prepare_text(self, text):
        """Host half of PromptLearner.forward for the CURRENT task (prompt_learner.py:128-133: "X"*n_ctx + caption + "." -> BPE ids), plus the packed row
        layout of the text tower — everything that can run ahead of the step on a worker thread (lpi_amd.pipeline).  -> PackedIds | LongTensor (host)."""
        ids = self.classifier_pool[self.numtask - 1](text)
        if not ids.is_cuda and self.args.get("trim_text", True):
            from lpi_amd.engine import PackedIds, trim_token_ids
            return PackedIds(ids, self._shared_rows()) if self.args.get("pack_text", True) else trim_token_ids(ids).contiguous()
        return ids

    def task_factors(self):
        """The five factors of the task being trained, in DecomposedPromptFn's argument order."""
        pr = self.prompts[self.numtask - 1]
        return {k: getattr(pr, k) for k in synth.PROMPT_NAMES}

    def train_step(self, image, text, flat_grad=None, grad_views=None, marks=None):
        """net(image, text) -> cal_loss -> sum(losses) -> backward (sprompt.py:303-310) as one fused call: leaves the gradients in the task's five factors
        (.grad; slices of flat_grad when given: optim.flatten) and returns {'loss': {base_loss, alignment_loss[, task_loss]}} like cal_loss.  `text`: what
        forward takes, or what prepare_text returned (already on the device)."""
        from lpi_amd.engine import PackedIds
        from lpi_amd.step import train_step
        eng = self._ensure_engine()
        ids = text if isinstance(text, PackedIds) or (torch.is_tensor(text) and text.is_cuda) else self._ids(text, self.numtask - 1, self._shared_rows())
        if isinstance(ids, PackedIds):
            ids.to(eng.device)
        term = None
        if self.numtask != 1:
            if self._task_term is None or self._task_term[0] != self.numtask:
                self._task_term = (self.numtask, _TaskTerm(self, self.numtask - 1))
            term = self._task_term[1]
        out = train_step(eng, image, ids, self.task_factors(), self.depth, self.exchange, flat_grad=flat_grad, grad_views=grad_views, task_term=term,
                         marks=marks)
        losses = {"base_loss": out["base_loss"], "alignment_loss": out["alignment_loss"]}
        if term is not None:
            losses["task_loss"] = out["task_loss"]      # (visual part, textual part): summed where the value is read (no kernel in the step)
        return {"loss": losses, "image_features": out["img_f"], "text_features": out["txt_f"]}

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

    @staticmethod
    def _task_target(task_id):
        path = "./MID/task_sim_matrix.txt"                          # cwd-relative like slinet.py:171
        if not os.path.exists(path):
            path = os.path.join(os.path.dirname(_HERE), "MID", "task_sim_matrix.txt")
        sim = torch.tensor(np.loadtxt(path)[:task_id + 1, :task_id + 1])
        return (sim > 0.4).type(torch.int)

    def cal_task_loss(self, task_id, visual_prompt, textual_prompt):
        dev = self.prompts[0].dim_1_share.device
        target = self._task_target(task_id).to(dev)
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
        self._task_term = None      # caches the finished tasks' stacks for ONE numtask

    def copy(self):
        """Deep copy of the trainable state; the frozen engine (operand weights + workspace arena) and the frozen f32 masters (clip_model) are shared, not
        duplicated (the reference deep-copies all 149.78 M parameters after every task, slinet.py:226-227)."""
        eng, clip, term = self.engine, self._modules.pop("clip_model"), self._task_term
        self.engine = self._task_term = None
        try:
            new = copy.deepcopy(self)
        finally:
            self.engine, self._task_term = eng, term
            self._modules["clip_model"] = clip
        new.engine = eng
        new._modules["clip_model"] = clip
        return new

    def load_state_dict(self, state_dict, *args, **kwargs):
        """nn.Module.load_state_dict; when the state carries backbone tensors (clip_model.*) the engine's operand copies (bf16 / fp16 / transposed) are stale:
        the engine is dropped and rebuilt from the new masters at the next use."""
        out = super().load_state_dict(state_dict, *args, **kwargs)
        if any(k.startswith("clip_model.") for k in state_dict):
            self.engine = None
        if any(k.startswith(("clip_model.", "prompts.")) for k in state_dict):
            self._task_term = None      # it caches the reconstructed stacks of the finished tasks: stale once any factor is replaced
        return out

    def trainable_state_dict(self):
        """The state a checkpoint of a continual run needs besides the frozen backbone: the 12 x 5 prompt factors (+ numtask)."""
        sd = {k: v.detach().clone() for k, v in self.state_dict().items() if k.startswith("prompts.")}
        sd["numtask"] = torch.tensor(self.numtask)
        return sd

    def load_trainable_state_dict(self, sd):
        sd = dict(sd)
        self.numtask = int(sd.pop("numtask", self.numtask))
        own = self.state_dict()
        unknown = [k for k in sd if k not in own or not k.startswith("prompts.")]
        if unknown:
            raise KeyError(f"not prompt factors of this network: {unknown[:3]}")
        with torch.no_grad():
            for k, v in sd.items():
                own[k].copy_(v)          # in place: FlatSGD's seating and the parameters' device stay as they are
        self._task_term = None           # the fused task term caches the finished tasks' stacks (rows 0..t-1): rebuilt from the restored factors
        return self

    def freeze(self):
        for param in self.parameters():
            param.requires_grad = False
        self.eval()
        return self


def stack_device(net):
    return net.prompts[0].dim_1_share.device

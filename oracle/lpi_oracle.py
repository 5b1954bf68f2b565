"""ORACLE — test infrastructure, NOT product code.

CPU restatement (plain tensor arithmetic on torch CPU tensors, float32 or float64) of the reference's
retrieval hot path: DecomposedPrompt -> prompted CLIP ViT / text transformer -> contrastive losses ->
prompt gradients, plus the evaluation scoring.  Every function cites the reference lines it follows
(paths relative to /root/reference/retrieval/).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; the product (``lpi_amd``) never does.

Pinning: the reference has no tests or golden vectors of its own, so this oracle is
pinned against outputs of the reference itself, captured by importing it in the build container
(``tests/golden/gen_golden.py`` -> ``tests/golden/*.npz``; checked by ``tests/test_oracle_golden.py``).
Third-party arithmetic underneath the reference is PyTorch ATen (version unpinned by the reference;
fixtures were generated with torch 2.10.0 CPU fp32).

``prompt_depth`` is a real parameter here.  depth=1 is what the shipped reference computes (its deep
prompt guard ``layer_id != 0 and layer_id < 0`` is dead, model.py:190 — SURVEY.md F1); depth>1 follows the
intended rule ``0 < layer_id < depth: x[1:P+1] += prompts[:, layer_id]`` (model.py:191-193) and is pinned
only by "patched" fixtures.
"""
from __future__ import annotations

import math

import numpy as np
import torch

LN_EPS = 1e-5  # nn.LayerNorm default, model.py:154-160

# Precision-mode EMULATION (tests/test_precision_modes.py only; None = exact arithmetic in the oracle's dtype, the parity oracle).
# OPERAND_DTYPE: every matrix-product operand of the two towers is rounded to this type first (products and sums stay in the oracle's
# dtype: what an MFMA with f32 accumulation does).  STREAM_DTYPE: the residual stream is rounded to this type after every block
# (the HIP path's fp16 stream in bf16 mode).  Answers "what does the operand type cost" without any kernel.
OPERAND_DTYPE = None
STREAM_DTYPE = None


def _op(t):
    return t if OPERAND_DTYPE is None else t.to(OPERAND_DTYPE).to(t.dtype)


def _stream(t):
    return t if STREAM_DTYPE is None else t.to(STREAM_DTYPE).to(t.dtype)


# ----------------------------------------------------------------------------- element ops
def layer_norm(x, w, b):
    """model.py:154-160 (fp32 LayerNorm, eps 1e-5, biased variance)."""
    mu = x.mean(-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(-1, keepdim=True)
    return xc * torch.rsqrt(var + LN_EPS) * w + b


def quick_gelu(x):
    """model.py:163-165."""
    return x * torch.sigmoid(1.702 * x)


def l2_normalise(x):
    """slinet.py:122,133: x / x.norm(dim=-1, keepdim=True) (no eps)."""
    return x / torch.sqrt((x * x).sum(-1, keepdim=True))


def decomposed_prompt(f, scale=1.0):
    """prompts.py:38-57: mean over r of d1[l,r]*d2[p,r]*d3[d,r]; visual and textual share dim_1_share."""
    d1 = f["dim_1_share"]
    r = d1.shape[1]
    vis = torch.einsum("lr,pr,dr->lpd", d1, f["dim_2_visual"], f["dim_3_visual"]) / r * scale
    txt = torch.einsum("lr,pr,dr->lpd", d1, f["dim_2_textual"], f["dim_3_textual"]) / r * scale
    return vis, txt


def interact(p, visual_out, textual_out, layer_id, a=0.1, eps=1e-5):
    """InteractModule.forward (grounding/maskrcnn_benchmark/modeling/bert/modeling_bert.py:616-651; optional row (f4) of SURVEY section 8):
    per direction M = mean_r d1[l, r] d2[i, r] d3[j, r] with a bias row (:617-627, :631-640), both new rows from the ORIGINAL inputs, then
    LayerNorm((1 - a) x + a x_new) per modality (:641-645; a = 0.1, norm_flag True, scale 1).  p: dict with the module's parameter names."""
    def m(d1, d2, d3):
        return torch.einsum("r,ir,jr->ij", d1[layer_id], d2, d3) / d1.shape[1]
    Dv, Dt = visual_out.shape[-1], textual_out.shape[-1]
    m_v2t = m(p["dim_1_v2t"], p["dim_2_v2t"], p["dim_3_v2t"])          # [Dv + 1, Dt]
    m_t2v = m(p["dim_1_t2v"], p["dim_2_t2v"], p["dim_3_t2v"])          # [Dt + 1, Dv]
    t_new = visual_out @ m_v2t[:Dv] + m_v2t[Dv:]
    v_new = textual_out @ m_t2v[:Dt] + m_t2v[Dt:]
    v = layer_norm((1 - a) * visual_out + a * v_new, p["visual_norm.weight"], p["visual_norm.bias"])
    t = layer_norm((1 - a) * textual_out + a * t_new, p["textual_norm.weight"], p["textual_norm.bias"])
    return v, t


# ----------------------------------------------------------------------------- transformer
def attention(x, w_in, b_in, w_out, b_out, heads, causal):
    """nn.MultiheadAttention as called at model.py:183-185, batch-first here ([B, L, d]).

    q is scaled by head_dim**-0.5 before q.k^T (F.multi_head_attention_forward); additive -inf mask above
    the diagonal for the text tower (model.py:347-353)."""
    B, L, d = x.shape
    hd = d // heads
    qkv = _op(_op(x) @ _op(w_in).t() + b_in)      # (stored in the operand type)
    q, k, v = qkv.split(d, dim=-1)
    q = q.reshape(B, L, heads, hd).transpose(1, 2) * (hd ** -0.5)
    k = k.reshape(B, L, heads, hd).transpose(1, 2)
    v = v.reshape(B, L, heads, hd).transpose(1, 2)
    s = _op(q) @ _op(k).transpose(-1, -2)
    if causal:
        s = s + torch.full((L, L), float("-inf"), dtype=x.dtype).triu_(1)
    p = torch.softmax(s, dim=-1)
    o = (_op(p) @ _op(v)).transpose(1, 2).reshape(B, L, d)
    return _op(o) @ _op(w_out).t() + b_out


def res_block(x, W, pre, heads, causal):
    """model.py:194-195."""
    x = x + attention(layer_norm(x, W[pre + "ln_1.weight"], W[pre + "ln_1.bias"]),
                      W[pre + "attn.in_proj_weight"], W[pre + "attn.in_proj_bias"],
                      W[pre + "attn.out_proj.weight"], W[pre + "attn.out_proj.bias"], heads, causal)
    x = _stream(x)
    h = layer_norm(x, W[pre + "ln_2.weight"], W[pre + "ln_2.bias"])
    h = quick_gelu(_op(h) @ _op(W[pre + "mlp.c_fc.weight"]).t() + W[pre + "mlp.c_fc.bias"])
    return _stream(x + _op(h) @ _op(W[pre + "mlp.c_proj.weight"]).t() + W[pre + "mlp.c_proj.bias"])


def transformer(x, W, prefix, layers, heads, causal, prompts, depth):
    """model.py:199-207 with the deep-prompt rule of model.py:189-193 (see module docstring for depth)."""
    for i in range(layers):
        if prompts is not None and 0 < i < depth:
            P = prompts.shape[-2]
            x = torch.cat([x[:, :1], x[:, 1:P + 1] + prompts[:, i], x[:, P + 1:]], dim=1)
        x = res_block(x, W, f"{prefix}resblocks.{i}.", heads, causal)
    return x


class Oracle:
    def __init__(self, cfg, state_dict, dtype=torch.float32):
        self.cfg = cfg
        self.dtype = dtype
        self.W = {k: torch.as_tensor(np.asarray(v)).to(dtype) for k, v in state_dict.items()}

    # ------------------------------------------------------------------ vision tower
    def encode_image(self, image, prompts=None, depth=1):
        """VisionTransformer.forward, model.py:227-259.  prompts: [B, Lyr, P, Dv] or None."""
        W, c = self.W, self.cfg
        B = image.shape[0]
        ps, g = c.vision_patch_size, c.image_resolution // c.vision_patch_size
        # conv1 (stride = kernel = patch, no bias) == per-patch matmul, model.py:228-230
        patches = image.to(self.dtype).reshape(B, 3, g, ps, g, ps).permute(0, 2, 4, 1, 3, 5).reshape(B, g * g, 3 * ps * ps)
        x = patches @ W["visual.conv1.weight"].reshape(c.vision_width, -1).t()
        cls = W["visual.class_embedding"].expand(B, 1, -1)
        x = torch.cat([cls, x], dim=1) + W["visual.positional_embedding"]          # model.py:235,243
        if prompts is not None:                                                       # model.py:240-248 (no pos-emb on prompts)
            x = torch.cat([x[:, :1], prompts[:, 0].to(self.dtype).expand(B, -1, -1), x[:, 1:]], dim=1)
        x = layer_norm(x, W["visual.ln_pre.weight"], W["visual.ln_pre.bias"])
        x = transformer(x, W, "visual.transformer.", c.vision_layers, c.vision_heads, False, prompts, depth)
        x = layer_norm(x[:, 0], W["visual.ln_post.weight"], W["visual.ln_post.bias"])
        return x @ W["visual.proj"]

    # ------------------------------------------------------------------ text tower
    def text_embed(self, ids, ctx=None, n_ctx=16):
        """PromptLearner.forward, CLASS_TOKEN_POSITION == 'end' (prompt_learner.py:128-163): embed ids
        (no grad), overwrite positions 1..n_ctx with ctx.  ctx None -> extract_vector (:118-126)."""
        emb = self.W["token_embedding.weight"][ids]
        if ctx is None:
            return emb
        ctx = ctx.to(self.dtype)
        if ctx.dim() == 2:
            ctx = ctx.unsqueeze(0).expand(ids.shape[0], -1, -1)
        return torch.cat([emb[:, :1], ctx, emb[:, 1 + n_ctx:]], dim=1)

    def encode_text(self, embeds, ids, prompts=None, depth=1):
        """TextEncoder.forward, prompt_learner.py:52-63."""
        W, c = self.W, self.cfg
        # The reference adds all context_length rows (prompt_learner.py:53); slicing to the sequence given is the identity at the
        # reference's own length and lets tests/ state the dead-row property (columns behind every EOT change nothing).
        x = embeds + W["positional_embedding"][: embeds.shape[1]]
        x = transformer(x, W, "transformer.", c.transformer_layers, c.transformer_heads, True, prompts, depth)
        x = layer_norm(x, W["ln_final.weight"], W["ln_final.bias"])
        x = x[torch.arange(x.shape[0]), ids.argmax(dim=-1)]
        return x @ W["text_projection"]

    # ------------------------------------------------------------------ SliNet surface
    def forward(self, image, ids, factors, depth=1):
        """SliNet.forward, slinet.py:109-135 (prompt_type == 'lpi')."""
        vis, txt = decomposed_prompt(factors)
        B = image.shape[0]
        vp = vis.unsqueeze(0).expand(B, -1, -1, -1)
        tp = txt.unsqueeze(0).expand(B, -1, -1, -1)
        img_f = l2_normalise(self.encode_image(image, vp, depth))
        emb = self.text_embed(ids, tp[:, 0])
        txt_f = l2_normalise(self.encode_text(emb, ids, tp, depth))
        return img_f, txt_f, vp, tp

    def extract_vector(self, image):
        """slinet.py:94-101 (un-prompted)."""
        return l2_normalise(self.encode_image(image, None))

    def extract_textual_vector(self, ids):
        """slinet.py:103-107 (raw placeholder embeddings, no ctx)."""
        return l2_normalise(self.encode_text(self.text_embed(ids, None), ids, None))

    def visual_interface(self, image, sel, all_factors, depth=1):
        """slinet.py:212-220: per-sample prompts gathered over tasks."""
        stack = torch.stack([decomposed_prompt(f)[0] for f in all_factors], 0)[sel]
        return l2_normalise(self.encode_image(image, stack, depth))

    def textual_interface(self, ids, sel, all_factors, depth=1):
        """slinet.py:185-210 (eval branch): per-caption task prompt."""
        stack = torch.stack([decomposed_prompt(f)[1] for f in all_factors], 0)[sel]
        emb = self.text_embed(ids, stack[:, 0])
        return l2_normalise(self.encode_text(emb, ids, stack, depth))

    def cal_loss(self, img_f, txt_f, vp, tp, numtask=1, all_factors=None, task_sim=None):
        """SliNet.cal_loss, slinet.py:137-165."""
        logits = self.W["logit_scale"].exp() * img_f @ txt_f.t()
        losses = {"base_loss": clip_loss(logits)}
        v = vp.mean(-1)
        t = tp.mean(-1)
        if v.dim() == 3:
            v, t = v.mean(0), t.mean(0)
        v, t = v / 0.01, t / 0.01
        losses["alignment_loss"] = 0.1 * clip_loss(v @ t.t())
        if numtask != 1:
            losses["task_loss"] = 0.1 * task_loss(numtask - 1, all_factors, task_sim)
        return losses, logits


# ----------------------------------------------------------------------------- losses
def clip_loss(logits):
    """ClipLoss.forward, loss/loss.py:75-87: (CE(logits, arange) + CE(logits^T, arange)) / 2."""
    n = logits.shape[0]
    idx = torch.arange(n)
    lr = torch.logsumexp(logits, dim=1) - logits[idx, idx]
    lc = torch.logsumexp(logits, dim=0) - logits[idx, idx]
    return (lr.mean() + lc.mean()) / 2


def nt_bxent_loss(x, target, temperature):
    """loss/loss.py:6-33 (as written: BCE-with-logits applied to an already-sigmoided input)."""
    n = x.shape[0]
    target = target.to(torch.float32)
    xn = x / torch.clamp(torch.sqrt((x * x).sum(-1, keepdim=True)), min=1e-8)   # F.cosine_similarity eps
    xcs = xn @ xn.t()
    xcs = xcs.masked_fill(torch.eye(n, dtype=torch.bool), float("inf"))
    z = torch.sigmoid(xcs / temperature)
    # binary_cross_entropy_with_logits(z, t) = max(z,0) - z*t + log(1 + exp(-|z|))
    loss = torch.clamp(z, min=0) - z * target + torch.log1p(torch.exp(-z.abs()))
    pos = target.bool()
    loss_pos = torch.where(pos, loss, torch.zeros_like(loss)).sum(1)
    loss_neg = torch.where(~pos, loss, torch.zeros_like(loss)).sum(1)
    num_pos = target.sum(1)
    num_neg = n - num_pos
    return (loss_pos / num_pos + loss_neg / num_neg).mean()


def task_loss(task_id, all_factors, task_sim):
    """SliNet.cal_task_loss, slinet.py:167-183: threshold 0.4 on MID/task_sim_matrix.txt, T = 0.001."""
    tgt = (torch.as_tensor(task_sim[:task_id + 1, :task_id + 1]) > 0.4).to(torch.int)
    vs = torch.stack([decomposed_prompt(all_factors[i])[0].reshape(-1) for i in range(task_id + 1)])
    ts = torch.stack([decomposed_prompt(all_factors[i])[1].reshape(-1) for i in range(task_id + 1)])
    return (nt_bxent_loss(vs, tgt, 0.001) + nt_bxent_loss(ts, tgt, 0.001)) / 2


# ----------------------------------------------------------------------------- train step
def train_step(oracle: Oracle, image, ids, factors_np, depth=1, numtask=1, all_factors_np=None, task_sim=None):
    """One forward + cal_loss + backward as SPrompts.train_function does (sprompt.py:297-311).

    Returns a dict of numpy outputs incl. the five prompt-factor gradients."""
    dt = oracle.dtype
    if all_factors_np is None:
        all_factors_np = [factors_np]
    allf = [{k: torch.as_tensor(v).to(dt) for k, v in f.items()} for f in all_factors_np]
    fac = allf[numtask - 1]
    for v in fac.values():
        v.requires_grad_(True)
    image = torch.as_tensor(image).to(dt)
    ids = torch.as_tensor(ids)
    img_f, txt_f, vp, tp = oracle.forward(image, ids, fac, depth)
    losses, logits = oracle.cal_loss(img_f, txt_f, vp, tp, numtask, allf, task_sim)
    total = sum(losses.values())
    total.backward()
    out = {"img_f": img_f, "txt_f": txt_f, "logits": logits, "vis_prompt": vp[0], "txt_prompt": tp[0]}
    out.update(losses)
    res = {k: v.detach().numpy() for k, v in out.items()}
    for k, v in fac.items():
        res["grad." + k] = v.grad.detach().numpy()
    return res


# ----------------------------------------------------------------------------- evaluation
def task_id_by_keys(feature, keys):
    """SPrompts.get_visual_task_id / get_textual_task_id, sprompt.py:336-368: L1 distance
    (written ((f-c)**2)**0.5 summed) to each task's centres; min over centres, argmin over tasks."""
    d = (feature[:, None, None, :] - keys[None]).abs().sum(-1)      # [B, T, C]
    return d.min(dim=2)[0].min(dim=1)[1]


def itm_eval(scores_i2t, scores_t2i, txt2img, img2txt, category_i, category_t, task_num):
    """SPrompts.itm_eval, sprompt.py:550-646: per-row descending argsort, best ground-truth rank,
    R@1/5/10 per task."""
    ranks = np.zeros(scores_i2t.shape[0])
    for index, score in enumerate(scores_i2t):
        inds = np.argsort(score)[::-1]
        ranks[index] = min(np.where(inds == i)[0][0] for i in img2txt[index])
    category_i = np.asarray(category_i)
    category_t = np.asarray(category_t)
    i2t, t2i = {}, {}
    for task in range(task_num):
        r = ranks[category_i == task]
        i2t[task] = [100.0 * (r < k).sum() / len(r) for k in (1, 5, 10)]
    ranks = np.zeros(scores_t2i.shape[0])
    for index, score in enumerate(scores_t2i):
        inds = np.argsort(score)[::-1]
        ranks[index] = np.where(inds == txt2img[index])[0][0]
    for task in range(task_num):
        r = ranks[category_t == task]
        t2i[task] = [100.0 * (r < k).sum() / len(r) for k in (1, 5, 10)]
    return {"mscoco": {"i2t": i2t, "t2i": t2i}}


# ----------------------------------------------------------------------------- task keys (KMeans)
def kmeans_fit(X, n_clusters=5, random_state=0, max_iter=300, tol=1e-4):
    """The fit behind the reference's task keys — ``KMeans(n_clusters=5, random_state=0).fit(features)`` (methods/sprompt.py:393-394) — restated in numpy.
    The algorithm is a third-party dependency that is not under /root/reference: scikit-learn (unpinned by the reference; 1.7.2 in the build
    container), `sklearn/cluster/_kmeans.py`: KMeans.fit with init='k-means++', n_init='auto' (= 1), algorithm='lloyd' —
      * the data are centred (X -= X.mean(0)) and the centres shifted back at the end; tol = mean(var(X, 0)) * 1e-4;
      * k-means++ seeding with 2 + int(log k) local trials per centre, driven by numpy's RandomState(random_state): choice(n, p=uniform), then per centre
        uniform(size=trials) * potential -> searchsorted in the float64 cumulative sum of the closest squared distances -> the candidate with the
        smallest new potential; squared distances evaluated in float64 and rounded to float32 (sklearn's float32 path upcasts in chunks);
      * Lloyd iterations: labels = argmin_c (|c|^2 - 2 x.c) in float32, centres = means of their points, until the labels repeat (strict convergence)
        or sum_c |c_new - c_old|^2 <= tol, then (if not strictly converged) one more labelling pass.
    Pinned by tests/golden/kmeans.npz (the imported reference's clustering() on synthetic features).  Test infrastructure only.
    X: [n, E] array-like (float32) -> (centers [k, E] float32, labels [n] int32, n_iter)."""
    import numpy as np
    X = np.array(X, dtype=np.float32, order="C", copy=True)
    n = X.shape[0]
    tol_ = np.mean(np.var(X, axis=0)) * tol
    rs = np.random.RandomState(random_state)
    w = np.ones(n, dtype=np.float32)
    X_mean = X.mean(axis=0)
    X -= X_mean

    def sqdist(C):      # [c, n] float32: exact differences in float64, rounded once
        d = ((C.astype(np.float64)[:, None, :] - X.astype(np.float64)[None, :, :]) ** 2).sum(-1)
        return np.maximum(d, 0).astype(np.float32)

    trials = 2 + int(np.log(n_clusters))
    centers = np.empty((n_clusters, X.shape[1]), dtype=np.float32)
    cid = rs.choice(n, p=w / w.sum())
    centers[0] = X[cid]
    closest = sqdist(centers[0:1])
    pot = closest @ w
    for c in range(1, n_clusters):
        rand_vals = rs.uniform(size=trials) * pot
        cand = np.searchsorted(np.cumsum(w * closest, dtype=np.float64).ravel(), rand_vals.ravel())
        np.clip(cand, None, n - 1, out=cand)
        dc = sqdist(X[cand])
        np.minimum(closest, dc, out=dc)
        pots = dc @ w.reshape(-1, 1)
        best = int(np.argmin(pots))
        pot, closest = pots[best], dc[best:best + 1]
        centers[c] = X[cand[best]]
    labels, labels_old = np.full(n, -1, np.int32), np.full(n, -2, np.int32)
    strict = False
    for it in range(max_iter):
        d = (centers * centers).sum(1)[None, :] - 2.0 * (X @ centers.T)          # float32, as the chunked Cython loop
        labels = d.argmin(1).astype(np.int32)
        new = np.zeros_like(centers)
        cnt = np.zeros(n_clusters, dtype=np.float32)
        np.add.at(new, labels, X)
        np.add.at(cnt, labels, 1.0)
        if (cnt == 0).any():
            # sklearn/cluster/_k_means_common.pyx _relocate_empty_clusters_dense: every empty cluster takes the point farthest from its own (old) centre
            # (the n_empty largest squared distances, in np.argpartition's order), its donor gives the point up; labels are left for the next iteration
            empty = np.where(cnt == 0)[0]
            dist = ((X - centers[labels]) ** 2).sum(axis=1)
            far = np.argpartition(dist, -len(empty))[:-len(empty) - 1:-1]
            for j, new_id in enumerate(empty):
                old_id = labels[far[j]]
                new[old_id] -= X[far[j]]
                new[new_id] = X[far[j]]
                cnt[new_id] = 1.0
                cnt[old_id] -= 1.0
        new /= np.where(cnt > 0, cnt, 1.0)[:, None]
        shift = np.sqrt(((new - centers) ** 2).sum(1))
        centers = new
        if np.array_equal(labels, labels_old):
            strict = True
            break
        if (shift ** 2).sum() <= tol_:
            break
        labels_old = labels.copy()
    if not strict:
        d = (centers * centers).sum(1)[None, :] - 2.0 * (X @ centers.T)
        labels = d.argmin(1).astype(np.int32)
    return (centers + X_mean).astype(np.float32), labels, it + 1

"""torch.autograd glue over the HIP engine: each Function's forward/backward only enqueues kernels of
``liblpi_hip.so`` (via ``engine``), so ``loss.backward()`` in the reference's training loop
(methods/sprompt.py:308-311) drives the hand-written backward.  No arithmetic of the path runs in ATen.
"""
from __future__ import annotations

import torch

from . import engine as E


def _own(dpr, ws):
    """The prompt gradient an encoder backward returned is the tower's PERSISTENT workspace buffer (engine.DualEncoder._dprompts), overwritten by the
    engine's next backward: autograd (hooks, torch.autograd.grad, retained .grad tensors) gets a private copy.  The fused step (step.train_step) consumes
    the buffer at once and sets ws['dprompts_borrow'] to skip the copy: no ATen kernel on its path."""
    if dpr is None or ws.get("dprompts_borrow", False):
        return dpr
    return dpr.clone()


class DecomposedPromptFn(torch.autograd.Function):
    """DecomposedPrompt.forward (models/prompts/prompts.py:38-57): (vis [Lyr,P,Dv], txt [Lyr,P,Dt])."""

    @staticmethod
    def forward(ctx, d1, d2v, d2t, d3v, d3t, scale=1.0, grad_out=None):
        """grad_out: five contiguous f32 tensors (in argument order) that receive the gradients — slices of a flat gradient buffer
        (optim.flatten), so that the parameters' .grad are views of it and the optimiser / the all-reduce run on one flat vector."""
        args = [t.detach().contiguous().float() for t in (d1, d2v, d2t, d3v, d3t)]
        ctx.save_for_backward(*args)
        ctx.scale, ctx.grad_out = scale, grad_out
        return E.prompt_cp_fwd2(args[0], args[1], args[2], args[3], args[4], scale)

    @staticmethod
    def backward(ctx, gvis, gtxt):
        d1, d2v, d2t, d3v, d3t = ctx.saved_tensors
        go = ctx.grad_out
        if gvis is None:
            gvis = torch.zeros(d1.shape[0], d2v.shape[0], d3v.shape[0], device=d1.device)
        if gtxt is None:
            gtxt = torch.zeros(d1.shape[0], d2t.shape[0], d3t.shape[0], device=d1.device)
        # both stacks in two launches; dim_1_share's gradient is the visual contribution plus the textual one, in that order
        g1, g2v, g2t, g3v, g3t = E.prompt_cp_bwd2(d1, d2v, d2t, d3v, d3t, gvis.float(), gtxt.float(), ctx.scale, out=go)
        if go is not None:      # fresh view objects: autograd's AccumulateGrad adopts a gradient it holds the only reference to, and clones one it does not
            g1, g2v, g2t, g3v, g3t = (t.view(t.shape) for t in (g1, g2v, g2t, g3v, g3t))
        return g1, g2v, g2t, g3v, g3t, None, None


class EncodeImageFn(torch.autograd.Function):
    """image_encoder(image, prompts) followed by the L2 normalisation of slinet.py:121-122."""

    @staticmethod
    def forward(ctx, enc, image, prompts, depth):
        ctx.enc = enc
        ctx.pshape = prompts.shape
        train = prompts.requires_grad
        # the backward context (workspace arena + the tower's forward serial) lives on THIS node: a later forward on the same engine
        # makes encode_image_backward raise instead of differentiating the wrong batch
        out, ctx.lpi = enc.encode_image(image, prompts.detach(), depth, train=train, return_ctx=True)
        # a fresh alias: the context holds `out` itself, and a tensor that is BOTH this node's output (grad_fn -> node) and held by the node's
        # context is a reference cycle through a C++ object — Python's collector cannot break it, and the engine with its workspace arena
        # (tens of GB) would never be freed
        return out.detach()

    @staticmethod
    def backward(ctx, g):
        dpr = _own(ctx.enc.encode_image_backward(g, ctx.lpi), ctx.lpi[0])
        if len(ctx.pshape) == 4:     # stride-0 expanded [B,Lyr,P,d] view: autograd sums the broadcast itself
            full = torch.zeros(ctx.pshape, device=dpr.device)
            full[0] = dpr
            dpr = full
        return None, None, dpr, None


class EncodeTextFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, enc, ids, prompts, depth):
        ctx.enc = enc
        ctx.pshape = prompts.shape
        train = prompts.requires_grad
        out, ctx.lpi = enc.encode_text(ids, prompts.detach(), depth, train=train, return_ctx=True)
        return out.detach()          # see EncodeImageFn.forward

    @staticmethod
    def backward(ctx, g):
        dpr = _own(ctx.enc.encode_text_backward(g, ctx.lpi), ctx.lpi[0])
        if len(ctx.pshape) == 4:
            full = torch.zeros(ctx.pshape, device=dpr.device)
            full[0] = dpr
            dpr = full
        return None, None, dpr, None


class EncodeBothFn(torch.autograd.Function):
    """EncodeImageFn and EncodeTextFn as ONE node: the two independent towers (slinet.py:121-133) run in lock step so that their GEMMs
    of the same layer op go out as one grouped launch (engine.DualEncoder.encode_both), forward and backward.  Same kernels, same bits
    as the two separate nodes."""

    @staticmethod
    def forward(ctx, enc, image, ids, vis_prompts, txt_prompts, depth):
        ctx.enc = enc
        ctx.vshape, ctx.tshape = vis_prompts.shape, txt_prompts.shape
        train = vis_prompts.requires_grad or txt_prompts.requires_grad
        (img_f, ctx.lpi_v), (txt_f, ctx.lpi_t) = enc.encode_both(image, ids, vis_prompts.detach(), txt_prompts.detach(), depth, train=train)
        return img_f.detach(), txt_f.detach()          # fresh aliases: see EncodeImageFn.forward

    @staticmethod
    def backward(ctx, g_img, g_txt):
        if g_img is None:
            g_img = torch.zeros_like(ctx.lpi_v[7])
        if g_txt is None:
            g_txt = torch.zeros_like(ctx.lpi_t[7])
        dv, dt = ctx.enc.encode_both_backward(g_img, g_txt, ctx.lpi_v, ctx.lpi_t)
        dv, dt = _own(dv, ctx.lpi_v[0]), _own(dt, ctx.lpi_t[0])
        outs = []
        for dpr, shape in ((dv, ctx.vshape), (dt, ctx.tshape)):
            if len(shape) == 4:     # stride-0 expanded [B,Lyr,P,d] view: autograd sums the broadcast itself
                full = torch.zeros(shape, device=dpr.device)
                full[0] = dpr
                dpr = full
            outs.append(dpr)
        return None, None, None, outs[0], outs[1], None


def contrastive_loss_and_grads(img_f, txt_f, scale, gather=None, exchange=None, need=True):
    """The contrastive loss of this rank and its gradients w.r.t. the rank's own feature rows, in the mode the exchange selects (see
    ClipLossFn) -> (loss [1], dI [B,E] | None, dT | None, logits | None).  Shared by ClipLossFn and step.train_step (which seeds the
    towers' backward with dI / dT directly)."""
    if gather is not None:
        img_all, txt_all, r0 = gather(img_f.detach(), txt_f.detach())
    else:
        img_all, txt_all, r0 = img_f.detach().contiguous(), txt_f.detach().contiguous(), 0
    B = img_f.shape[0]
    ll = bool(getattr(exchange, "local_loss", False))
    gwg = bool(getattr(exchange, "gather_with_grad", False))
    logits = None
    if ll:
        loss, dI, dT, dIk, dTk = E.clip_loss_local_fwd_bwd(img_all, txt_all, scale, r0, B, need, key_grads=gwg and need)
        if need and gwg:
            kI, kT = exchange.reduce_scatter_rows(dIk, dTk, B)        # this rank's rows of the SUM over ranks
            dI, dT = dI + kI, dT + kT
    elif gwg:
        loss, dIa, dTa = E.clip_loss_full_grad(img_all, txt_all, scale)
        dI, dT = exchange.reduce_scatter_rows(dIa, dTa, B) if need else (None, None)
    else:
        loss, logits, dI, dT = E.clip_loss_fwd_bwd(img_all, txt_all, scale, need, r0, B)     # gradients of the local rows only
    return loss, dI, dT, logits


class ClipLossFn(torch.autograd.Function):
    """logit_scale * I @ T^T then ClipLoss (slinet.py:138-141, loss/loss.py:75-87).  With a process group the features are
    all-gathered first (dp.py) in one of the four modes of the reference's dead ``gather_features`` / ``get_logits`` /
    ``ClipLoss.get_ground_truth`` (sprompt.py:38-82, 272-288; loss/loss.py:62-73), selected on the ``dp.Exchange``:

    * local_loss=False, gather_with_grad=False (default): every rank evaluates the FULL global loss and back-propagates only through
      its own rows (sprompt.py:75-80); no backward collective.  SUM of the ranks' parameter gradients = the global-batch gradient.
    * local_loss=True: the rank's images against all texts and its texts against all images, labels offset by rank * B; the value is the
      rank's own mean.  Without gather_with_grad the gathered features carry no gradient (the reference's partial gradient); with it the
      key gradients are reduce-scattered (SUM) to their owners, as ``torch.distributed.nn.all_gather``'s backward does.
    * local_loss=False, gather_with_grad=True: the full loss with gradients through every gathered feature, reduce-scattered.
    In the last three modes the mean over ranks of the parameter gradients is the quantity of interest, so step.py weighs the loss by
    1/W before the SUM all-reduce."""

    @staticmethod
    def forward(ctx, img_f, txt_f, scale, gather=None, exchange=None):
        need = img_f.requires_grad or txt_f.requires_grad
        loss, dI, dT, ctx.logits = contrastive_loss_and_grads(img_f, txt_f, scale, gather, exchange, need)
        if need:
            ctx.save_for_backward(dI, dT)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        dI, dT = ctx.saved_tensors
        return dI * g, dT * g, None, None, None


class AlignLossFn(torch.autograd.Function):
    """0.1 * ClipLoss((mean_d vis / 0.01) @ (mean_d txt / 0.01)^T)   (slinet.py:143-158)."""

    @staticmethod
    def forward(ctx, vis, txt, temp, weight):
        need = vis.requires_grad or txt.requires_grad
        loss, dv, dt = E.align_loss_fwd_bwd(vis.detach().float(), txt.detach().float(), temp, weight, need)
        if need:
            ctx.save_for_backward(dv, dt)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        dv, dt = ctx.saved_tensors
        return dv * g, dt * g, None, None


class NtBxentFn(torch.autograd.Function):
    """nt_bxent_loss over the stacked, flattened prompts of tasks 0..t (loss/loss.py:6-33 via slinet.py:167-183).  Only row `row`
    (the task being trained) receives a gradient: the other tasks' prompts are frozen (sprompt.py:230-237)."""

    @staticmethod
    def forward(ctx, X, target, temp, row):
        Xc = X.detach().contiguous().float()
        T, D = Xc.shape
        loss = torch.empty(1, device=X.device)
        need = X.requires_grad
        dx = torch.empty(D, device=X.device) if need else None
        scratch = torch.empty(2 * T * T, device=X.device)
        E.call("lpi_nt_bxent_fwd_bwd", T, D, int(row) if need else -1, Xc, target.to(device=X.device, dtype=torch.int32).contiguous(),
               float(temp), 1.0, loss, dx, 0, scratch, E._stream())
        ctx.row, ctx.shape = int(row), (T, D)
        if need:
            ctx.save_for_backward(dx)
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        (dx,) = ctx.saved_tensors
        full = torch.zeros(ctx.shape, device=dx.device)
        full[ctx.row] = dx * g
        return full, None, None, None


class InteractFn(torch.autograd.Function):
    """InteractModule.forward (grounding/maskrcnn_benchmark/modeling/bert/modeling_bert.py:616-651): the low-rank cross-modal mix of the visual and
    textual prompt rows followed by the two LayerNorms — lpi_interact_fwd / lpi_interact_bwd.  Inputs [..., Dv] / [..., Dt] with equal leading
    dimensions; parameters in the module's order."""

    @staticmethod
    def forward(ctx, visual_out, textual_out, layer_id, d1v, d2v, d3v, d1t, d2t, d3t, gv, bv, gt, bt, mix=0.1, eps=1e-5):
        lead = visual_out.shape[:-1]
        if textual_out.shape[:-1] != lead:
            raise ValueError("visual and textual prompt rows must have the same leading dimensions")
        Dv, Dt = visual_out.shape[-1], textual_out.shape[-1]
        xv = visual_out.detach().reshape(-1, Dv).contiguous().float()
        xt = textual_out.detach().reshape(-1, Dt).contiguous().float()
        N, (Lyr, R) = xv.shape[0], d1v.shape
        ps = [t.detach().contiguous().float() for t in (d1v, d2v, d3v, d1t, d2t, d3t, gv, bv, gt, bt)]
        if ps[1].shape != (Dv + 1, R) or ps[2].shape != (Dt, R) or ps[4].shape != (Dt + 1, R) or ps[5].shape != (Dv, R) or ps[3].shape != (Lyr, R):
            raise ValueError("InteractModule factor shapes do not match the inputs")
        ov, ot = torch.empty_like(xv), torch.empty_like(xt)
        stat = torch.empty(4, N, device=xv.device)
        E.call("lpi_interact_fwd", N, Dv, Dt, R, Lyr, int(layer_id), xv, Dv, xt, Dt, *ps, float(mix), float(eps), ov, Dv, ot, Dt, stat, E._stream())
        ctx.save_for_backward(xv, xt, *ps)
        ctx.meta = (N, Dv, Dt, R, Lyr, int(layer_id), float(mix), float(eps), lead)
        return ov.reshape(*lead, Dv), ot.reshape(*lead, Dt)

    @staticmethod
    def backward(ctx, g_v, g_t):
        xv, xt, *ps = ctx.saved_tensors
        N, Dv, Dt, R, Lyr, layer, mix, eps, lead = ctx.meta
        gv_ = (torch.zeros_like(xv) if g_v is None else g_v.reshape(-1, Dv).contiguous().float())
        gt_ = (torch.zeros_like(xt) if g_t is None else g_t.reshape(-1, Dt).contiguous().float())
        dxv, dxt = torch.empty_like(xv), torch.empty_like(xt)
        la, lb = Dt * R + (Dv + 1) * R + R + 2 * Dt, Dv * R + (Dt + 1) * R + R + 2 * Dv
        grads = torch.empty(la + lb, device=xv.device)
        nws = E._lib.load().lpi_interact_workspace_floats(N, Dv, Dt, R)
        wsp = torch.empty(nws, device=xv.device)
        E.call("lpi_interact_bwd", N, Dv, Dt, R, Lyr, layer, xv, Dv, xt, Dt, *ps, mix, eps, gv_, Dv, gt_, Dt, dxv, Dv, dxt, Dt, grads, wsp, E._stream())

        def split(v, Din, Dout):
            o = 0
            d3 = v[o:o + Dout * R].view(Dout, R); o += Dout * R
            d2 = v[o:o + (Din + 1) * R].view(Din + 1, R); o += (Din + 1) * R
            d1row = v[o:o + R]; o += R
            dg = v[o:o + Dout]; o += Dout
            db = v[o:o + Dout]
            d1 = torch.zeros(Lyr, R, device=v.device)
            d1[layer] = d1row
            return d1, d2, d3, dg, db
        d1v, d2v, d3v, dgt, dbt = split(grads[:la], Dv, Dt)          # v2t: its LayerNorm is the textual one
        d1t, d2t, d3t, dgv, dbv = split(grads[la:], Dt, Dv)
        return (dxv.reshape(*lead, Dv), dxt.reshape(*lead, Dt), None, d1v, d2v, d3v, d1t, d2t, d3t, dgv, dbv, dgt, dbt, None, None)

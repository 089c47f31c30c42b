"""Coarse autograd nodes gluing the explicit HIP forward/backward sequences into torch.autograd, so that the
SPADE-lineage trainer code (`loss.backward()`) works unchanged.

Every network is ONE node; parameter gradients never travel through autograd: the wgrad kernels accumulate them
directly into the network's flat grad buffer (ParamStore.grad).  `anchor` is a dummy 1-element tensor with
requires_grad=True whose only job is to make autograd call the node's backward.
"""
import torch

from .. import ops
from .._lib import chunk_elems


OVERLAP_VGG = True      # VGG branch of the G loss on a side stream (tools/ab_step.py flips it for an A/B)


def _vgg_stream(model):
    s = getattr(model, "_vgg_side", None)
    if s is None:
        s = model._vgg_side = torch.cuda.Stream()
    return s


def _anchor(net):
    a = getattr(net, "_anchor", None)
    if a is None or a.device != net.store.master.device:
        a = torch.zeros(1, device=net.store.master.device, requires_grad=True)
        net._anchor = a
    return a


class _GeneratorNode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, prev_image, state, net, save):
        dt = net.compute_dtype
        img = ops.nchw_to_nhwc(prev_image.float(), dt, chunk_elems(dt))
        out, c = net.fwd_nhwc(img, state.float(), save=save)
        ctx.net, ctx.c = net, c
        return out

    @staticmethod
    def backward(ctx, d_out):
        if not ctx.c:
            raise RuntimeError("generator forward was run without saving activations")
        ctx.net.bwd_nhwc(ctx.c, d_out.contiguous())
        ctx.c = None
        return None, None, None, None, None


def generator_apply(net, prev_image, state):
    """NHWC compute-dtype output [N,H,W,ce] (3 real channels)."""
    save = torch.is_grad_enabled()
    return _GeneratorNode.apply(_anchor(net) if save else _anchor(net).detach(), prev_image, state, net, save)


class _NhwcToNchw(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, C):
        ctx.dt, ctx.pitch = x.dtype, x.shape[3]
        return ops.nhwc_to_nchw(x, C)

    @staticmethod
    def backward(ctx, g):
        return ops.nchw_to_nhwc(g.contiguous().float(), ctx.dt, ctx.pitch), None


def nhwc_to_nchw_apply(x, C):
    return _NhwcToNchw.apply(x, C)


# ---------------------------------------------------------------------------------------------------------
def _build_d_input(model, fake_nhwc, prev_image, real_image):
    """cat([prev;fake] , [prev;real]) along batch, channels 0..2 = prev_image, 3..5 = image, pitch 8."""
    dt = model.netD.compute_dtype
    N, _, H, W = prev_image.shape
    x = torch.empty((2 * N, H, W, 8), dtype=dt, device=prev_image.device)
    ops.nchw_to_nhwc(prev_image, dt, 8, out=x[:N], c_off=0, zero_pad=True)
    ops.nchw_to_nhwc(prev_image, dt, 8, out=x[N:], c_off=0, zero_pad=True)
    ops.copy_channels(fake_nhwc, 0, x, 3, 3, src_rows=N)
    ops.nchw_to_nhwc(real_image, dt, 8, out=x[N:], c_off=3, zero_pad=False)
    return x


def _hinge_seed(logits, mode_lo, mode_hi, N, num_D, loss_lo, loss_hi):
    """logits: NHWC [2N,h,w,ce] (1 real channel).  Applies hinge mode_lo to the first N samples and mode_hi to the
    last N (if not None); returns the gradient in the layout of `logits` ([N,...] only when mode_hi is None)."""
    B, h, w, ce = logits.shape
    keep = B if mode_hi is not None else N
    dense = ops.nhwc_to_nchw(logits[:keep], 1)              # [keep,1,h,w] fp32
    gd = torch.empty_like(dense)
    cnt = N * h * w
    sc = 1.0 / (cnt * num_D)
    ops.hinge_loss(dense, cnt, mode_lo, sc, loss_lo, gd, x_off=0)
    if mode_hi is not None:
        ops.hinge_loss(dense, cnt, mode_hi, sc, loss_hi, gd, x_off=cnt)
    return ops.nchw_to_nhwc(gd, logits.dtype, ce)


class _GLossNode(torch.autograd.Function):
    """G-step losses: hinge GAN + feature matching (through netD, frozen) + VGG perceptual + pixel L1."""

    @staticmethod
    def forward(ctx, fake, model, prev_image, real_image):
        opt = model.opt
        N, H, W, ce = fake.shape
        dt = fake.dtype
        if model.before_netD is not None:
            # data-parallel hook (a StepGraph cut point): D's weight update of the previous step must have landed before
            # netD is used below.  It sits here, ahead of the stream fork, because a graph segment cannot end while a
            # forked side stream has not re-joined.
            model.before_netD()
        losses = torch.zeros(4, dtype=torch.float32, device=fake.device)
        # pixel L1 + VGG share the NHWC copy of the real image
        both = torch.empty((2 * N, H, W, ce), dtype=dt, device=fake.device)
        both[:N].copy_(fake)
        ops.nchw_to_nhwc(real_image, dt, ce, out=both[N:])
        d_fake = torch.zeros_like(fake)
        if opt.lambda_l1 > 0:
            ops.l1_loss(both[:N], both[N:], opt.lambda_l1 / (N * 3 * H * W), losses[3:4], d_fake)
        # The perceptual (VGG) branch and the discriminator branch are independent given `fake`: VGG runs on a side stream,
        # concurrently with D (the fork / join is captured by hipGraph); two MFMA-bound chains of mid-sized launches fill each
        # other's tails and inter-kernel gaps.
        main = torch.cuda.current_stream()
        vgg_on_side = (not opt.no_vgg_loss) and OVERLAP_VGG
        vs = _vgg_stream(model) if vgg_on_side else main
        vctx, tap_grads = None, None
        if not opt.no_vgg_loss:
            from .networks.loss import VGG_WEIGHTS
            if vgg_on_side:
                vs.wait_stream(main)
                both.record_stream(vs); losses.record_stream(vs)
            with torch.cuda.stream(vs):
                taps, vctx = model.vgg.fwd_nhwc(both)
                tap_grads = []
                for wk, t in zip(VGG_WEIGHTS, taps):
                    tg = torch.empty_like(t[:N])
                    ops.l1_loss(t[:N], t[N:], opt.lambda_vgg * wk / t[:N].numel(), losses[2:3], tg)
                    tap_grads.append(tg)
        x = _build_d_input(model, fake, prev_image, real_image)
        res, dctx = model.netD.fwd_nhwc(x)
        num_D = len(res)
        grads = []
        for feats in res:
            g = [None] * len(feats)
            g[-1] = _hinge_seed(feats[-1], 2, None, N, num_D, losses[0:1], None)
            if not opt.no_ganFeat_loss:
                for j in range(len(feats) - 1):
                    f = feats[j]
                    gf = torch.empty_like(f[:N])             # gradient of the fake half only (real half: detached)
                    half = f[:N].numel()
                    ops.l1_loss(f[:N], f[N:], opt.lambda_feat / num_D / half, losses[1:2], gf)
                    g[j] = gf
            grads.append(g)
        if vgg_on_side:
            main.wait_stream(vs)
        ctx.vgg_on_side = vgg_on_side
        ctx.model, ctx.dctx, ctx.grads, ctx.vctx, ctx.tap_grads, ctx.d_fake, ctx.N = \
            model, dctx, grads, vctx, tap_grads, d_fake, N
        # one zero-dim output per loss term (not one [4] tensor indexed by the caller): the trainer's
        # sum(losses.values()).mean().backward() then costs no select-backward zero-fill / copy / accumulate kernels
        return tuple(losses.unbind(0))

    @staticmethod
    def backward(ctx, *gs_in):
        model, N = ctx.model, ctx.N
        if not model.assume_unit_loss_grad:
            # general case: scale each loss's gradient seeds by its upstream gradient (device scalars, no host sync)
            g = torch.stack([gi.float() if gi is not None else torch.zeros((), device=ctx.d_fake.device) for gi in gs_in])
            for gs in ctx.grads:
                ops.scale_(gs[-1], g[0:1])
                for t in gs[:-1]:
                    if t is not None:
                        ops.scale_(t, g[1:2])
            if ctx.tap_grads is not None:
                for t in ctx.tap_grads:
                    ops.scale_(t, g[2:3])
            ops.scale_(ctx.d_fake, g[3:4])
        d_fake = ctx.d_fake
        main = torch.cuda.current_stream()
        dv = None
        if ctx.vctx is not None:                # VGG backward on its side stream, under the discriminator backward
            vs = _vgg_stream(model) if ctx.vgg_on_side else main
            if ctx.vgg_on_side:
                vs.wait_stream(main)
            with torch.cuda.stream(vs):
                dv = model.vgg.bwd_nhwc(ctx.vctx, ctx.tap_grads, N)
            if ctx.vgg_on_side:
                dv.record_stream(main)
        dx = model.netD.bwd_nhwc(ctx.dctx, ctx.grads, need_wgrad=False, need_dx=True, n_keep=N)   # fake half only
        ops.copy_channels(dx, 3, d_fake, 0, 3, accumulate=True, src_rows=N)
        if dv is not None:
            if ctx.vgg_on_side:
                main.wait_stream(vs)
            ops.add(d_fake, dv, out=d_fake)
        ctx.dctx = ctx.grads = ctx.vctx = ctx.tap_grads = None
        return d_fake, None, None, None


def g_losses_apply(model, fake_nhwc, prev_image, real_image):
    return _GLossNode.apply(fake_nhwc, model, prev_image, real_image)


class _DLossNode(torch.autograd.Function):
    """D-step losses: hinge on D(prev, fake.detach()) and D(prev, real); weight grads go to netD's flat buffer."""

    @staticmethod
    def forward(ctx, anchor, model, fake, prev_image, real_image):
        N = fake.shape[0]
        losses = torch.zeros(2, dtype=torch.float32, device=fake.device)
        x = _build_d_input(model, fake, prev_image, real_image)
        if model.before_netD is not None:
            model.before_netD()
        res, dctx = model.netD.fwd_nhwc(x)
        num_D = len(res)
        grads = []
        for feats in res:
            g = [None] * len(feats)
            g[-1] = _hinge_seed(feats[-1], 0, 1, N, num_D, losses[0:1], losses[1:2])
            grads.append(g)
        ctx.model, ctx.dctx, ctx.grads, ctx.N = model, dctx, grads, N
        return tuple(losses.unbind(0))

    @staticmethod
    def backward(ctx, *gs_in):
        model = ctx.model
        dev = ctx.grads[0][-1].device
        if not model.assume_unit_loss_grad:
            g = torch.stack([gi.float() if gi is not None else torch.zeros((), device=dev) for gi in gs_in])
            for gs in ctx.grads:
                t = gs[-1]
                half = t.numel() // 2
                flat = t.view(-1)
                ops.scale_(flat[:half], g[0:1])
                ops.scale_(flat[half:], g[1:2])
        model.netD.bwd_nhwc(ctx.dctx, ctx.grads, need_wgrad=True, need_dx=False)
        ctx.dctx = ctx.grads = None
        return None, None, None, None, None


def d_losses_apply(model, fake_nhwc, prev_image, real_image):
    return _DLossNode.apply(_anchor(model.netD), model, fake_nhwc.detach(), prev_image, real_image)

"""Coarse autograd nodes gluing the explicit HIP forward/backward sequences into torch.autograd, so that the
SPADE-lineage trainer code (`loss.backward()`) works unchanged.

Every network is ONE node; parameter gradients never travel through autograd: the wgrad kernels accumulate them
directly into the network's flat grad buffer (ParamStore.grad).  `anchor` is a dummy 1-element tensor with
requires_grad=True whose only job is to make autograd call the node's backward.
"""
import torch

from .. import ops, stepgraph
from .._lib import chunk_elems


DBWD_LANE = 0           # discriminator backward: coarser scales on this side-stream lane under scale 0 (None: one stream)
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
def _build_d_half(model, prev_image, image_nchw=None, image_nhwc=None):
    """One half of the D batch (the reference's cat([fake; real]) along the batch is run as two passes of N):
    cat(prev_image, image) on channels (0..2 | 3..5), pitch 8; image given as fp32 NCHW (real) or as the generator's
    NHWC output (fake)."""
    dt = model.netD.compute_dtype
    N, _, H, W = prev_image.shape
    x = torch.empty((N, H, W, 8), dtype=dt, device=prev_image.device)
    ops.nchw_to_nhwc(prev_image, dt, 8, out=x, c_off=0, zero_pad=True)
    if image_nhwc is not None:
        ops.copy_channels(image_nhwc, 0, x, 3, 3, src_rows=N)
    else:
        ops.nchw_to_nhwc(image_nchw, dt, 8, out=x, c_off=3, zero_pad=False)
    return x


def _side(model, name):
    s = getattr(model, name, None)
    if s is None:
        s = torch.cuda.Stream()
        setattr(model, name, s)
    return s


def vgg_real_prefetch(model, real_image, dt, ce):
    """G step: the perceptual features of the REAL image depend on nothing the step computes, so they are taken on the VGG
    side stream while the generator forward runs on the main stream.  Returns the handle _GLossNode consumes."""
    main = torch.cuda.current_stream()
    vs = _vgg_stream(model)
    stepgraph.fork(vs, main)
    real_image.record_stream(vs)
    with torch.cuda.stream(vs):
        real_nhwc = ops.nchw_to_nhwc(real_image, dt, ce)
        taps, acts = model.vgg.fwd_nhwc(real_nhwc)
    return dict(real_nhwc=real_nhwc, taps=taps, acts=acts, stream=vs)


OVERLAP_DREAL = True    # the real half of the discriminator batch on a side stream (G step: under the generator forward)
DREAL_EARLY_BWD = True  # D step: its backward too, ahead of the fake half (only when the losses' upstream gradients are 1)
DREAL_REUSE = True      # D step: take netD(prev, real) from the G step of the same train step instead of recomputing it


def _dreal_key(model, prev_image, real_image):
    """What D(prev, real) depends on: the two input tensors (identity + in-place version) and netD's packed weights."""
    return (id(prev_image), prev_image._version, prev_image.data_ptr(), id(real_image), real_image._version,
            real_image.data_ptr(), getattr(model.netD.store, "version", 0))


def _record_ctx(dctx, stream):
    for x, saved in dctx:
        x.record_stream(stream)
        for tup in saved:
            for t in tup:
                if t is not None:
                    t.record_stream(stream)


def dreal_pass(model, prev_image, real_image, on_side):
    """Discriminator forward on (prev, real) -- the half of the D batch that depends on nothing either step computes.
    The G step needs its features (feature matching), the D step its activations (weight gradients): netD's weights do
    not change between the two, so the pass is done ONCE per train step, here, and the D step picks it up from
    model._dreal_cache (_DStepNode).  on_side: on the `_dreal_side` stream (the caller joins `stream` before using it)."""
    main = torch.cuda.current_stream()
    side = _side(model, "_dreal_side") if on_side else main
    if on_side:
        stepgraph.fork(side, main)
        prev_image.record_stream(side); real_image.record_stream(side)
    with torch.cuda.stream(side):
        xr = _build_d_half(model, prev_image, image_nchw=real_image)
        # (no nested fork: HIP's stream capture crashed in hipStreamEndCapture with a side stream forked from a side stream)
        res, dctx = model.netD.fwd_nhwc(xr, lane=None if on_side else 0)
    if on_side:
        _record_ctx(dctx, main)
    return dict(res=res, dctx=dctx, stream=side if on_side else None, key=_dreal_key(model, prev_image, real_image))


def _hinge_seed(logits, mode, N, num_D, loss_slot):
    """logits: NHWC [N,h,w,ce] (1 real channel).  loss_slot += hinge `mode` averaged over the map and over the num_D scales;
    returns the gradient in the layout of `logits` (one launch, straight on the NHWC map)."""
    B, h, w, ce = logits.shape
    return ops.hinge_loss_nhwc(logits, mode, 1.0 / (N * h * w * num_D), loss_slot)


class _GLossNode(torch.autograd.Function):
    """G-step losses: hinge GAN + feature matching (through netD, frozen) + VGG perceptual + pixel L1."""

    @staticmethod
    def forward(ctx, fake, model, prev_image, real_image, pre, pre_d):
        opt = model.opt
        N, H, W, ce = fake.shape
        dt = fake.dtype
        main = torch.cuda.current_stream()
        if pre is not None:
            main.wait_stream(pre["stream"])     # VGG(real) prefetch re-joins here (it ran under the generator forward)
        if pre_d is not None and pre_d["stream"] is not None:
            main.wait_stream(pre_d["stream"])   # ... and so does the D(prev, real) pass
        if model.before_netD is not None:
            # data-parallel hook (a StepGraph cut point): D's weight update of the previous step must have landed before
            # netD is used below.  It sits here, ahead of the stream fork, because a graph segment cannot end while a
            # forked side stream has not re-joined.
            model.before_netD()
        losses = torch.zeros(4, dtype=torch.float32, device=fake.device)
        real_nhwc = pre["real_nhwc"] if pre is not None else ops.nchw_to_nhwc(real_image, dt, ce)
        d_fake = torch.zeros_like(fake)
        if opt.lambda_l1 > 0:
            ops.l1_loss(fake, real_nhwc, opt.lambda_l1 / (N * 3 * H * W), losses[3:4], d_fake)
        # The perceptual (VGG) branch and the discriminator branch are independent given `fake`: VGG runs on a side stream,
        # concurrently with D (the fork / join is captured by hipGraph); two MFMA-bound chains of mid-sized launches fill each
        # other's tails and inter-kernel gaps.  VGG sees the fake and the real image as two batches of N: the real one
        # may already have been done under the generator forward (vgg_real_prefetch).
        vgg_on_side = (not opt.no_vgg_loss) and OVERLAP_VGG and not ops.SERIALIZE
        vs = _vgg_stream(model) if vgg_on_side else main
        vctx, tap_grads, vctx_real = None, None, None
        if not opt.no_vgg_loss:
            from .networks.loss import VGG_WEIGHTS
            if vgg_on_side:
                stepgraph.fork(vs, main)
                fake.record_stream(vs); losses.record_stream(vs); real_nhwc.record_stream(vs)
            with torch.cuda.stream(vs):
                if pre is not None:
                    taps_r, vctx_real = pre["taps"], pre["acts"]
                else:
                    taps_r, vctx_real = model.vgg.fwd_nhwc(real_nhwc)
                taps, vctx = model.vgg.fwd_nhwc(fake)
                tap_grads = [torch.empty_like(t) for t in taps]
                ops.l1_loss_multi([(t, tr, opt.lambda_vgg * wk / t.numel(), losses[2:3], tg)      # the 5 taps: one launch
                                   for wk, t, tr, tg in zip(VGG_WEIGHTS, taps, taps_r, tap_grads)])
        # The discriminator sees the two halves of its batch as two passes of N (InstanceNorm is per-sample, so the
        # halves are independent): (prev, real) may already have been done under the generator forward (dreal_pass in
        # compute_generator_loss); otherwise it goes on its side stream now, next to the (prev, fake) pass.
        dreal = pre_d
        if dreal is None:
            dreal = dreal_pass(model, prev_image, real_image, OVERLAP_DREAL and not ops.SERIALIZE)
        xf = _build_d_half(model, prev_image, image_nhwc=fake)
        res, dctx = model.netD.fwd_nhwc(xf)
        if pre_d is None and dreal["stream"] is not None:
            main.wait_stream(dreal["stream"])
        res_r = dreal["res"]
        model._dreal_cache = dreal                 # the D step of this train step reuses it (same inputs, same netD weights)
        num_D = len(res)
        grads, fm_jobs = [], []
        for feats, feats_r in zip(res, res_r):
            g = [None] * len(feats)
            g[-1] = _hinge_seed(feats[-1], 2, N, num_D, losses[0:1])
            if not opt.no_ganFeat_loss:
                for j in range(len(feats) - 1):
                    f = feats[j]
                    gf = torch.empty_like(f)                 # gradient of the fake pass only (real features: detached)
                    fm_jobs.append((f, feats_r[j], opt.lambda_feat / num_D / f.numel(), losses[1:2], gf))
                    g[j] = gf
            grads.append(g)
        if fm_jobs:
            ops.l1_loss_multi(fm_jobs)                       # the feature-matching maps of every scale: one launch
        if vgg_on_side:
            main.wait_stream(vs)
        ctx.vgg_on_side = vgg_on_side
        ctx.model, ctx.dctx, ctx.grads, ctx.vctx, ctx.tap_grads, ctx.d_fake, ctx.N = \
            model, dctx, grads, vctx, tap_grads, d_fake, N
        ctx.dctx_r = dreal["dctx"]                                                 # (parity tests' branch masks)
        ctx.vctx_real, ctx.real_nhwc, ctx.fake = vctx_real, real_nhwc, fake      # (kept for the parity tests' branch masks)
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
                stepgraph.fork(vs, main)
            with torch.cuda.stream(vs):
                dv = model.vgg.bwd_nhwc(ctx.vctx, ctx.tap_grads, N)
            if ctx.vgg_on_side:
                dv.record_stream(main)
        dx = model.netD.bwd_nhwc(ctx.dctx, ctx.grads, need_wgrad=False, need_dx=True, lane=DBWD_LANE)
        ops.copy_channels(dx, 3, d_fake, 0, 3, accumulate=True, src_rows=N)
        if dv is not None:
            if ctx.vgg_on_side:
                main.wait_stream(vs)
            ops.add(d_fake, dv, out=d_fake)
        ctx.dctx = ctx.dctx_r = ctx.grads = ctx.vctx = ctx.tap_grads = ctx.vctx_real = ctx.real_nhwc = ctx.fake = None
        return d_fake, None, None, None, None, None


def g_losses_apply(model, fake_nhwc, prev_image, real_image, pre=None, pre_d=None):
    return _GLossNode.apply(fake_nhwc, model, prev_image, real_image, pre, pre_d)


class _DStepNode(torch.autograd.Function):
    """The whole D step as one node: fake = G(prev, state) without gradient, hinge on D(prev, fake) and D(prev, real);
    weight gradients go to netD's flat buffer.  The two halves of the D batch are independent (InstanceNorm is
    per-sample), and the real half depends on nothing the step computes: its forward is normally the one the G step of the
    same train step already did (model._dreal_cache: same inputs, same netD weights -- the reference recomputes it);
    its backward runs on a side stream while the generator forward occupies the main stream when the losses' upstream
    gradients are known to be 1 (the trainer).  The fake half follows on the main stream."""

    @staticmethod
    def forward(ctx, anchor, model, prev_image, state, real_image, fake_given):
        N = prev_image.shape[0]
        netD, netG = model.netD, model.netG
        num_D = netD.num_D
        dev = prev_image.device
        losses = torch.zeros(2, dtype=torch.float32, device=dev)
        main = torch.cuda.current_stream()
        unit = model.assume_unit_loss_grad
        use_side = OVERLAP_DREAL and unit and not ops.SERIALIZE
        side = _side(model, "_dreal_side") if use_side else main

        def half(x, mode, loss_slot, lane):
            res, dctx = netD.fwd_nhwc(x, lane=lane)
            grads = []
            for feats in res:
                g = [None] * len(feats)
                g[-1] = _hinge_seed(feats[-1], mode, N, num_D, loss_slot)
                grads.append(g)
            return dctx, grads

        # the (prev, real) forward of this train step's G step, if it is still valid (same inputs, netD not updated since)
        cache = getattr(model, "_dreal_cache", None)
        model._dreal_cache = None
        if cache is not None and (not DREAL_REUSE or cache["key"] != _dreal_key(model, prev_image, real_image)):
            cache = None
        if use_side:
            stepgraph.fork(side, main)
            for t in (prev_image, real_image, losses):
                t.record_stream(side)
        with torch.cuda.stream(side):
            if cache is not None:
                dctx_r = cache["dctx"]
                grads_r = []
                for feats in cache["res"]:
                    g = [None] * len(feats)
                    g[-1] = _hinge_seed(feats[-1], 1, N, num_D, losses[1:2])
                    grads_r.append(g)
                cache = None
            else:
                xr = _build_d_half(model, prev_image, image_nchw=real_image)
                # (no nested fork: HIP's stream capture crashed in hipStreamEndCapture with a side stream forked from a side stream)
                dctx_r, grads_r = half(xr, 1, losses[1:2], None if use_side else 0)
            if unit and DREAL_EARLY_BWD:               # upstream gradient is 1: the real half's backward can go now
                netD.bwd_nhwc(dctx_r, grads_r, need_wgrad=True, need_dx=False, lane=None if use_side else DBWD_LANE)
                dctx_r = grads_r = None
        if fake_given is None:
            dt = netG.compute_dtype
            img = ops.nchw_to_nhwc(prev_image, dt, chunk_elems(dt))
            fake, _ = netG.fwd_nhwc(img, state, save=False)
        else:
            fake = fake_given
        if use_side:
            main.wait_stream(side)                     # D.grad is accumulated without atomics: the halves' backwards are ordered
        xf = _build_d_half(model, prev_image, image_nhwc=fake)
        dctx_f, grads_f = half(xf, 0, losses[0:1], 0)
        ctx.model, ctx.N = model, N
        ctx.dctx_f, ctx.grads_f, ctx.dctx_r, ctx.grads_r = dctx_f, grads_f, dctx_r, grads_r
        return tuple(losses.unbind(0))

    @staticmethod
    def backward(ctx, *gs_in):
        model = ctx.model
        netD = model.netD
        if not model.assume_unit_loss_grad:
            dev = ctx.grads_f[0][-1].device
            g = torch.stack([gi.float() if gi is not None else torch.zeros((), device=dev) for gi in gs_in])
            for gs in ctx.grads_f:
                ops.scale_(gs[-1], g[0:1])
            for gs in ctx.grads_r:
                ops.scale_(gs[-1], g[1:2])
        if ctx.dctx_r is not None:
            netD.bwd_nhwc(ctx.dctx_r, ctx.grads_r, need_wgrad=True, need_dx=False, lane=DBWD_LANE)
        netD.bwd_nhwc(ctx.dctx_f, ctx.grads_f, need_wgrad=True, need_dx=False, lane=DBWD_LANE)
        ctx.dctx_f = ctx.grads_f = ctx.dctx_r = ctx.grads_r = None
        return None, None, None, None, None, None


def d_step_apply(model, prev_image, state, real_image, fake_nhwc=None):
    return _DStepNode.apply(_anchor(model.netD), model, prev_image, state, real_image,
                            fake_nhwc.detach() if fake_nhwc is not None else None)

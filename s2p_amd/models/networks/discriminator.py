"""Multi-scale PatchGAN discriminator (`--netD multiscale`, sub-arch `n_layer`; SPADE lineage, README.md:73).

Input: NHWC tensor [B,H,W,8] holding cat(prev_image, image) in channels 0..5 (SPEC.md D7).  Each scale is a stack
of 4x4 convs (pad 2, strides 2,2,1 / last two stride 1), LeakyReLU(0.2), InstanceNorm on the middle layers; all
intermediate features are returned for the feature-matching loss.  Explicit forward/backward over HIP launches;
the feature-matching / hinge gradients of the intermediate features are folded into the dgrad epilogue (EPI_ADD).
"""
import os

import torch
import torch.nn as nn

from ... import ops, stepgraph
from ..._lib import ACT_LRELU, ACT_NONE, EPI_ADD, EPI_MUL_ACTGRAD, EPI_STORE
from ...ops import ConvGeom
from .base_network import BaseNetwork
from .generator import _Conv
from .layers import ConvLayer

LRELU = 0.2
DFWD_SIDE = True         # forward: coarser scales on a side stream under scale 0 (diagnostic switch)


class NLayerDiscriminator(BaseNetwork):
    @staticmethod
    def modify_commandline_options(parser, is_train):
        parser.add_argument("--n_layers_D", type=int, default=4, help="# layers in each discriminator")
        return parser

    def __init__(self, opt, input_nc=6):
        super().__init__()
        self.opt = opt
        self.n_layers = opt.n_layers_D
        self.input_nc = input_nc
        nf = opt.ndf
        self.model0 = _Conv(input_nc, nf, 4, bias=True)
        self.chans = [nf]
        for n in range(1, self.n_layers):
            nf_prev, nf = nf, min(nf * 2, 512)
            setattr(self, f"model{n}", _Conv(nf_prev, nf, 4, bias=False))      # InstanceNorm follows: no bias
            self.chans.append(nf)
        setattr(self, f"model{self.n_layers}", _Conv(nf, 1, 4, bias=True))
        self.chans.append(1)

    def _declare_packs(self, dt, store=None, prefix=""):
        st = store if store is not None else self.store
        self.lay = []
        cin = self.input_nc
        for n in range(self.n_layers + 1):
            m = getattr(self, f"model{n}")
            st.add(m.weight, "conv")
            if m.bias is not None:
                st.add(m.bias, "bias")
            stride = 2 if n < self.n_layers - 1 else 1
            pk = st.pack(f"{prefix}model{n}", [m.weight], [m.bias] if m.bias is not None else None, dtype=dt)
            self.lay.append(ConvLayer(pk, ConvGeom(cin, self.chans[n], 4, stride, 2, net="D")))
            cin = self.chans[n]

    def fwd_nhwc(self, x):
        """Returns (features list [f0..fn] NHWC, ctx)."""
        nl = self.n_layers
        f = self.lay[0].fwd(x, act=ACT_LRELU, slope=LRELU)
        feats, saved = [f], [(x, None, None, f)]
        for n in range(1, nl):
            xin = f
            # conv -> InstanceNorm -> LeakyReLU: ONE launch where a plane-resident kernel takes the layer (s2p_conv2d_fwd_mat)
            c, f, s = self.lay[n].fwd_mat(xin, None, 0, None, 0, ACT_LRELU, LRELU)
            feats.append(f)
            saved.append((xin, c, s, f))
        out = self.lay[nl].fwd(f)
        feats.append(out)
        saved.append((f, None, None, out))
        return feats, saved

    def bwd_nhwc(self, saved, grads, need_wgrad=True, need_dx=True, n_keep=None):
        """grads[j] = d(loss)/d(feature j) or None.  Returns d(loss)/dx (or None).  With `n_keep`, only the first
        n_keep samples are differentiated (InstanceNorm is per-sample, so a batch prefix is self-contained): the G step
        uses it because the real half of the D batch carries no gradient."""
        nl = self.n_layers
        if n_keep is not None:
            # (the statistics buffer is self-describing and valid for a batch prefix: it is not sliced)
            saved = [(xin[:n_keep], None if c is None else c[:n_keep], s, f[:n_keep]) for xin, c, s, f in saved]
        xin, _, _, out = saved[nl]
        g = grads[nl]
        if g is None:
            raise RuntimeError("discriminator backward needs a gradient for the final logit map")
        if need_wgrad:
            self.lay[nl].wgrad(xin, g)
        # dgrad of layer n + 1 -> (+ the tap gradient on feature n) -> backward of layer n's InstanceNorm + LeakyReLU: one call
        # (s2p_conv2d_dgrad_mat: ONE launch where a plane-resident kernel takes the dgrad), giving dL/d(conv_n output)
        dy, lay_next = g, self.lay[nl]
        for n in reversed(range(1, nl)):
            xin, c, s, f = saved[n]
            dc = lay_next.dgrad_mat(dy, c, s, None, 0, None, 0, ACT_LRELU, LRELU, None, 0, None, 0, aux=grads[n])
            if need_wgrad:
                self.lay[n].wgrad(xin, dc)
            dy, lay_next = dc, self.lay[n]
        # layer 1's dgrad: xin = f0 = LeakyReLU(conv0): fold the tap gradient and the LeakyReLU mask into the epilogue
        xin = saved[1][0]
        d = self.lay[1].dgrad(dy, xin.shape, aux=xin, aux2=grads[0], epi=EPI_MUL_ACTGRAD, aux_act=ACT_LRELU, slope=LRELU)
        x, _, _, f0 = saved[0]
        dpre = d
        if need_wgrad:
            self.lay[0].wgrad(x, dpre)
        if need_dx:
            return self.lay[0].dgrad(dpre, x.shape)
        return None


class MultiscaleDiscriminator(BaseNetwork):
    @staticmethod
    def modify_commandline_options(parser, is_train):
        parser.add_argument("--netD_subarch", type=str, default="n_layer", help="architecture of each discriminator")
        parser.add_argument("--num_D", type=int, default=2, help="number of discriminators (scales)")
        opt, _ = parser.parse_known_args()
        from . import find_network_using_name
        subnetD = find_network_using_name(opt.netD_subarch, "discriminator")
        subnetD.modify_commandline_options(parser, is_train)
        return parser

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.num_D = opt.num_D
        from . import find_network_using_name
        sub = find_network_using_name(getattr(opt, "netD_subarch", "n_layer"), "discriminator")
        for i in range(opt.num_D):
            self.add_module("discriminator_%d" % i, sub(opt))

    def subnets(self):
        return [getattr(self, "discriminator_%d" % i) for i in range(self.num_D)]

    def _declare_packs(self, dt):
        for i, d in enumerate(self.subnets()):
            d._declare_packs(dt, store=self.store, prefix=f"discriminator_{i}.")
            d.compute_dtype = dt
            d.finalized = True

    def _side_stream(self, lane=0):
        """Side stream for the coarser scales, one per `lane`: two concurrent passes (the D step runs the real half of its
        batch on its own stream, lane 1) must not serialise through a shared side stream."""
        d = self.__dict__.setdefault("_sides", {})
        if lane not in d:
            d[lane] = torch.cuda.Stream()
        return d[lane]

    def fwd_nhwc(self, x, lane=0):
        """The scales are independent given their inputs; the coarser scales' kernels are too small to fill 256 CUs, so
        scales >= 1 run on a side stream, overlapped with scale 0 (fork/join captured by hipGraph)."""
        self._require_ready()
        main = torch.cuda.current_stream()
        side = self._side_stream(lane) if (lane is not None and DFWD_SIDE and not ops.SERIALIZE) else main   # lane None: one stream
        subs = self.subnets()
        xs = [x]
        for i in range(1, self.num_D):
            xs.append(ops.avgpool_fwd(xs[-1]))
        result, ctx = [None] * self.num_D, [None] * self.num_D
        if side is not main:
            stepgraph.fork(side, main)
        with torch.cuda.stream(side):
            for i in range(1, self.num_D):
                if side is not main:
                    xs[i].record_stream(side)
                feats, saved = subs[i].fwd_nhwc(xs[i])
                if side is not main:
                    for f in feats:
                        f.record_stream(main)
                result[i], ctx[i] = feats, (xs[i], saved)
        feats, saved = subs[0].fwd_nhwc(xs[0])
        result[0], ctx[0] = feats, (xs[0], saved)
        if side is not main:
            main.wait_stream(side)
        return result, ctx

    def bwd_nhwc(self, ctx, grads, need_wgrad=True, need_dx=True, n_keep=None, lane=None):
        """Backward of every scale.  The scales write disjoint parameters and meet only in d(loss)/dx (the coarser scale's dx
        is folded into the finer one's through the adjoint of the average pool), so with `lane` given the scales >= 1 run on
        that lane's side stream under scale 0 -- same launches, same accumulation order as the one-stream form (lane None;
        used when the caller is itself on a side stream: no nested forks)."""
        main = torch.cuda.current_stream()
        fork = lane is not None and not ops.SERIALIZE and self.num_D > 1
        side = self._side_stream(lane) if fork else main
        subs = self.subnets()

        def one(i):
            x, saved = ctx[i]
            return subs[i].bwd_nhwc(saved, grads[i], need_wgrad=need_wgrad, need_dx=need_dx, n_keep=n_keep)

        def xshape(i):
            x = ctx[i][0]
            return tuple((x[:n_keep] if n_keep is not None else x).shape)

        dx_next = None
        if fork:
            stepgraph.fork(side, main)
            for i in range(1, self.num_D):                     # made on main or on a forward lane, read on `side`
                x, saved = ctx[i]
                for tup in saved:
                    for t in tup:
                        if t is not None and torch.is_tensor(t):
                            t.record_stream(side)
                for g in grads[i]:
                    if g is not None:
                        g.record_stream(side)
        with torch.cuda.stream(side):
            for i in reversed(range(1, self.num_D)):
                dx = one(i)
                if need_dx:
                    if dx_next is not None:
                        ops.avgpool_bwd(dx_next, xshape(i), dx=dx, accumulate=True)
                    dx_next = dx
            if fork and dx_next is not None:
                dx_next.record_stream(main)
        dx = one(0)
        if fork:
            main.wait_stream(side)
        if need_dx:
            if dx_next is not None:
                ops.avgpool_bwd(dx_next, xshape(0), dx=dx, accumulate=True)
            dx_next = dx
        return dx_next

    def forward(self, x):
        """x: fp32 NCHW [B,6,H,W] = cat(prev_image, image).  Returns List[List[Tensor]] (per scale, per layer) of fp32
        NCHW features, last entry = patch logits (inference/diagnostic API; the train step uses the fused loss nodes)."""
        self._require_ready()
        from ..._lib import chunk_elems
        with torch.no_grad():
            xd = ops.nchw_to_nhwc(x, self.compute_dtype, ops.pad_to(x.shape[1], chunk_elems(self.compute_dtype)))
            res, _ = self.fwd_nhwc(xd)
            out = []
            for d, feats in zip(self.subnets(), res):
                out.append([ops.nhwc_to_nchw(f, c) for f, c in zip(feats, d.chans)])
        return out

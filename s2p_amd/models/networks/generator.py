"""S2P generator, `--netG s2p` (README.md:33,59): (previous image, state) -> next image.

State-conditioned SPADE/MAT ResNet encoder-decoder (SPEC.md; rebuttal.md:146-154 for the MAT block).  The whole
network is ONE coarse autograd node whose forward/backward are explicit sequences of HIP launches; MI355X-first
restructuring relative to a per-layer eager graph:
  * the 12 MAT norms' image-conditioning branches depend only on `prev_image`, so their 12 `mlp_shared` convs run as
    ONE conv 3->12*128 and their 24 gamma/beta convs as ONE grouped conv (12 groups x 128->2C) that fills all 256 CUs;
  * the 12 state affines run as one fused linear 256 -> 12*2C;
  * InstanceNorm + modulation + LeakyReLU is one stats pass + one elementwise pass (s2p_in_*), residual adds are
    fused into the conv epilogue, weight gradients go straight into the flat grad buffer.
"""
import os

import torch
import torch.nn as nn

from ... import ops, stepgraph
from ..._lib import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_TANH, EPI_ADD, EPI_MUL_ACTGRAD, chunk_elems
from ...ops import ConvGeom, pad_to
from .base_network import BaseNetwork
from .layers import ConvLayer

LRELU = 0.2
WGRAD_SIDE = True        # deferred weight gradients on a side stream (A/B: tools/ab_step.py)
WGRAD_CHUNK_BLOCKS = 2   # ... launched every this many finished ResBlks, under the rest of the backward chain (0: one batch behind it)
STATE_SIDE_FWD = True    # state path on its side stream in the forward / in the backward (diagnostic switches)
STATE_SIDE_BWD = True
FUSE_SKIP_ADD = True     # ResBlk backward: the skip gradient is added inside the MAT backward launch (s2p_in_norm_bwd_res)
FUSE_NORM_FWD = True     # ResBlk forward: InstanceNorm + MAT + LeakyReLU in the epilogue of the producing conv (s2p_conv2d_fwd_mat)
FUSE_NORM_BWD = True     # ResBlk backward: each MAT norm's backward in the epilogue of the following dgrad (s2p_conv2d_dgrad_mat)
GB_GROUP_MAJOR = True    # the 12 gamma|beta planes (and their gradients) as [12][N][h][w][2C] (1-KB rows, one tensor per norm) instead of
                         # channel slices of one [N][h][w][12*2C] tensor (12-KB rows): the fused conv + norm tails read them 10 % faster
GB_WGRAD_CHUNKED = True  # with the data-parallel exchange on: the gamma/beta heads' weight gradients follow the ResBlk batches (4 norms per batch)
                         # instead of one 12-group launch at the end, so 2/3 of the gradient is in flight before the last weight-gradient launch.
                         # On one rank the single launch stays: same-box A/B 8.907 (one launch) vs 8.947 ms/step (three)
DEC_WGRAD_SIDE = False   # the output / up convs' weight gradients on the weight-gradient stream beside the decoder's backward chain.
                         # Off: same-box A/B 8.980 (side stream) vs 8.942 ms/step -- as in round 3, co-running kernels only trade CUs
COND_SIDE = True         # backward of the image-conditioning branch on its own stream, concurrent with the encoder backward
GB_FWD_CHUNKS = 1        # forward: the 12 gamma|beta planes produced in this many launches of 12 / n groups on the conditioning stream, each
                         # a chunk AHEAD of the ResBlk chain that consumes it (1: one 12-group launch the chain waits
                         # for).  Same-box A/B (tools/ab_step.py networks.generator.GB_FWD_CHUNKS): 8.783 (1) / 8.773 (2) / 8.772 (3) /
                         # 8.767 ms (6): nothing -- the chunks that run beside the chain take from it what the earlier start gives


class _Linear(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(cout, cin))
        self.bias = nn.Parameter(torch.zeros(cout))


class _Conv(nn.Module):
    def __init__(self, cin, cout, k, bias=True, transposed=False):
        super().__init__()
        shape = (cin, cout, k, k) if transposed else (cout, cin, k, k)
        self.weight = nn.Parameter(torch.zeros(shape))
        if bias:
            self.bias = nn.Parameter(torch.zeros(cout))
        else:
            self.bias = None


class StateMapping(nn.Module):
    """PositionalEncoding(L) + n_mlp x (Linear + LeakyReLU 0.2)  (README.md:74-75 lineage)."""

    def __init__(self, state_dim, L, w_dim, n_mlp):
        super().__init__()
        d = state_dim * (1 + 2 * L)
        for i in range(n_mlp):
            setattr(self, f"fc{i}", _Linear(d if i == 0 else w_dim, w_dim))


class MATNorm(nn.Module):
    """Parameters of one MAT norm: image branch (mlp_shared, mlp_gamma, mlp_beta) + state affine (fc_state)."""

    def __init__(self, norm_nc, nhidden, w_dim):
        super().__init__()
        self.mlp_shared = _Conv(3, nhidden, 3)
        self.mlp_gamma = _Conv(nhidden, norm_nc, 3)
        self.mlp_beta = _Conv(nhidden, norm_nc, 3)
        self.fc_state = _Linear(w_dim, 2 * norm_nc)


class MATResnetBlock(nn.Module):
    def __init__(self, c, nhidden, w_dim):
        super().__init__()
        self.norm_0 = MATNorm(c, nhidden, w_dim)
        self.norm_1 = MATNorm(c, nhidden, w_dim)
        self.conv_0 = _Conv(c, c, 3)
        self.conv_1 = _Conv(c, c, 3)


class S2PGenerator(BaseNetwork):
    @staticmethod
    def modify_commandline_options(parser, is_train):
        parser.add_argument("--resnet_n_downsample", type=int, default=2, help="number of stride-2 stages in G")
        parser.add_argument("--resnet_n_blocks", type=int, default=6, help="number of MAT residual blocks in G")
        parser.add_argument("--mat_nhidden", type=int, default=128, help="hidden width of the MAT image branch")
        parser.add_argument("--posenc_L", type=int, default=10, help="positional-encoding octaves for the state")
        parser.add_argument("--n_mlp", type=int, default=4, help="layers of the state mapping MLP")
        return parser

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.state_dim = opt.state_dim
        self.ngf, self.n_down, self.n_blocks = opt.ngf, opt.resnet_n_downsample, opt.resnet_n_blocks
        self.nhidden, self.w_dim, self.posenc_L, self.n_mlp = opt.mat_nhidden, opt.z_dim, opt.posenc_L, opt.n_mlp
        ngf = self.ngf
        self.state_map = StateMapping(self.state_dim, self.posenc_L, self.w_dim, self.n_mlp)
        self.stem = _Conv(3, ngf, 7, bias=False)
        c = ngf
        for i in range(self.n_down):
            setattr(self, f"down{i}", _Conv(c, 2 * c, 3, bias=False))
            c *= 2
        self.c_mid = c
        self.blocks = nn.ModuleList([MATResnetBlock(c, self.nhidden, self.w_dim) for _ in range(self.n_blocks)])
        for i in range(self.n_down):
            setattr(self, f"up{i}", _Conv(c, c // 2, 3, bias=False, transposed=True))
            c //= 2
        self.out = _Conv(c, 3, 7, bias=True)
        self.on_early_grads = None          # optional callable, see bwd_nhwc
        self.on_tail_final = None           # optional callable: runs (on the weight-gradient stream) once the early-complete tail of the
                                            # flat gradient is final -- the one-rank trainer applies that range's Adam update there

    @property
    def early_grad_offset(self):
        """Offset (elements) in the flat parameter / gradient buffers where the early-complete tail starts."""
        for e in self.store.entries:
            if e["param"] is self._early_first:
                return e["offset"]
        raise RuntimeError("network not finalized")

    def early_buckets(self):
        """The early-complete tail of the flat gradient as BUCKETS that become final one after the other during the backward
        (data-parallel exchange, Pix2PixTrainer): bucket j holds what the j-th batch of deferred weight gradients completes --
        the convs of WGRAD_CHUNK_BLOCKS ResBlks and the gamma/beta heads of their norms; the first bucket also holds the up / output
        convs, whose gradients are final from the start of the backward.  Each bucket is a list of (offset, numel) ranges."""
        off = {id(e["param"]): (e["offset"], e["numel"]) for e in self.store.entries}
        end = self.store.numel
        norms = self._norms()
        nb, step = self.n_blocks, (WGRAD_CHUNK_BLOCKS if WGRAD_CHUNK_BLOCKS > 0 else self.n_blocks)
        buckets = []
        for hi in range(nb - 1, -1, -step):
            lo = max(hi - step + 1, 0)
            n0, n1 = norms[2 * lo], norms[2 * hi + 1]
            r = [(off[id(n0.mlp_gamma.weight)][0], sum(off[id(n1.mlp_beta.weight)]) - off[id(n0.mlp_gamma.weight)][0]),
                 (off[id(n0.mlp_gamma.bias)][0], sum(off[id(n1.mlp_beta.bias)]) - off[id(n0.mlp_gamma.bias)][0])]
            c0 = off[id(self.blocks[lo].conv_0.weight)][0]
            c1 = end if hi == nb - 1 else sum(off[id(self.blocks[hi].conv_1.bias)])
            r.append((c0, c1 - c0))
            buckets.append(r)
        assert min(o for b in buckets for o, _ in b) == self.early_grad_offset
        assert sum(n for b in buckets for _, n in b) == end - self.early_grad_offset
        return buckets

    # ---- flat layout + packed operands ---------------------------------------------------------------------
    def _norms(self):
        return [n for b in self.blocks for n in (b.norm_0, b.norm_1)]

    def _declare_packs(self, dt):
        st, f32 = self.store, torch.float32
        C, nh = self.c_mid, self.nhidden
        norms = self._norms()
        nn_ = len(norms)
        L = {}
        # state path (always fp32: 0.6 MFLOP/img, latency-bound; keeps the conditioning exact)
        fcs = [getattr(self.state_map, f"fc{i}") for i in range(self.n_mlp)]
        for i, fc in enumerate(fcs):
            st.add(fc.weight, "linear"); st.add(fc.bias, "bias")
            pk = st.pack(f"state_map.fc{i}", [fc.weight], [fc.bias], kind="linear", dtype=f32, need_bwd=i > 0)
            L[f"fc{i}"] = ConvLayer(pk, ConvGeom(fc.weight.shape[1], fc.weight.shape[0], 1))
        for n in norms:
            st.add(n.fc_state.weight, "linear")
        for n in norms:
            st.add(n.fc_state.bias, "bias")
        pk = st.pack("fc_state_all", [n.fc_state.weight for n in norms], [n.fc_state.bias for n in norms],
                     kind="linear", dtype=f32)
        L["fc_state"] = ConvLayer(pk, ConvGeom(self.w_dim, nn_ * 2 * C, 1))
        # image-conditioning branch, batched over the 12 norms
        for n in norms:
            st.add(n.mlp_shared.weight, "conv")
        for n in norms:
            st.add(n.mlp_shared.bias, "bias")
        pk = st.pack("mlp_shared_all", [n.mlp_shared.weight for n in norms], [n.mlp_shared.bias for n in norms],
                     dtype=dt, need_bwd=False)
        L["shared"] = ConvLayer(pk, ConvGeom(3, nn_ * nh, 3, 1, 1))
        # encoder.  Flat-buffer order = declaration order, and it is chosen for the data-parallel gradient exchange: the
        # gradients that the backward finishes LAST (state path, shared conv, stem, down convs: 3 % of the parameters) come
        # first, everything that is complete once the ResBlk chain and the deferred weight gradients are done (gamma/beta
        # heads, ResBlk convs, up convs, output conv: 97 %) forms one contiguous tail -> ONE early all-reduce of that tail
        # overlaps the rest of the backward (Pix2PixTrainer / parallel.py).
        st.add(self.stem.weight, "conv")
        L["stem"] = ConvLayer(st.pack("stem", [self.stem.weight], dtype=dt, need_bwd=False),
                              ConvGeom(3, self.ngf, 7, 1, 3, reflect=True))
        c = self.ngf
        for i in range(self.n_down):
            d = getattr(self, f"down{i}")
            st.add(d.weight, "conv")
            L[f"down{i}"] = ConvLayer(st.pack(f"down{i}", [d.weight], dtype=dt), ConvGeom(c, 2 * c, 3, 2, 1))
            c *= 2
        self.__dict__["_early_first"] = norms[0].mlp_gamma.weight      # first parameter of the early-complete tail (not a module attribute: keeps state_dict clean)
        for n in norms:
            st.add(n.mlp_gamma.weight, "conv"); st.add(n.mlp_beta.weight, "conv")
        for n in norms:
            st.add(n.mlp_gamma.bias, "bias"); st.add(n.mlp_beta.bias, "bias")
        pk = st.pack("mlp_gb_all", [w for n in norms for w in (n.mlp_gamma.weight, n.mlp_beta.weight)],
                     [b for n in norms for b in (n.mlp_gamma.bias, n.mlp_beta.bias)], groups=nn_, dtype=dt)
        L["gb"] = ConvLayer(pk, ConvGeom(nh, 2 * C, 3, 1, 1, groups=nn_, x_gstride=nh, y_gstride=2 * C))
        for b, blk in enumerate(self.blocks):
            for j, cv in enumerate((blk.conv_0, blk.conv_1)):
                st.add(cv.weight, "conv"); st.add(cv.bias, "bias")
                L[f"b{b}c{j}"] = ConvLayer(st.pack(f"blocks.{b}.conv_{j}", [cv.weight], [cv.bias], dtype=dt),
                                           ConvGeom(c, c, 3, 1, 1))
        for i in range(self.n_down):
            u = getattr(self, f"up{i}")
            st.add(u.weight, "convT")
            L[f"up{i}"] = ConvLayer(st.pack(f"up{i}", [u.weight], kind="convT", dtype=dt),
                                    ConvGeom(c, c // 2, 3, 2, 1, transposed=True, output_padding=1))
            c //= 2
        st.add(self.out.weight, "conv"); st.add(self.out.bias, "bias")
        L["out"] = ConvLayer(st.pack("out", [self.out.weight], [self.out.bias], dtype=dt),
                             ConvGeom(c, 3, 7, 1, 3, reflect=True))
        for lay in L.values():
            lay.geom.net = "G"            # (bench.py's per-network aggregates: ops._Prof records it)
        self.lay = L

    # ---- explicit forward / backward on NHWC tensors -------------------------------------------------------
    def fwd_nhwc(self, img, state, save=True):
        """img: NHWC compute-dtype [N,H,W,ce] (3 real channels), state fp32 [N,S].  Returns (out NHWC, ctx)."""
        self._require_ready()
        L, C, nh = self.lay, self.c_mid, self.nhidden
        N, H, W, _ = img.shape
        if H % (1 << self.n_down) or W % (1 << self.n_down):
            raise ValueError(f"image size {H}x{W} must be a multiple of {1 << self.n_down}")
        ctx = {}
        # state path: a chain of tiny fp32 GEMMs (2..48 workgroups each).  It is independent of the image-conditioning
        # convs and the encoder, so it runs on a side stream and overlaps them (the fork/join is captured by hipGraph).
        main = torch.cuda.current_stream()
        side = self._side_stream() if STATE_SIDE_FWD else main
        stepgraph.fork(side, main)
        with torch.cuda.stream(side):
            # dedicated small-M fp32 kernels (csrc/linear_small.hip): one ~4 us launch per layer
            pe_dim = self.state_dim * (1 + 2 * self.L_oct)
            pe_pitch = pad_to(pe_dim, 4)
            h = ops.posenc(state.contiguous(), self.L_oct, pe_pitch)    # [N, pe_pitch]
            hs = [h]
            for i in range(self.n_mlp):
                pk = L[f"fc{i}"].pk
                h = ops.linear_fwd(h, pk.w_fwd, pk.bias, pk.Cpad, pk.R, ACT_LRELU, LRELU)
                hs.append(h)
            pk = L["fc_state"].pk
            st_all = ops.linear_fwd(h, pk.w_fwd, pk.bias, pk.Cpad, pk.R)  # [N, 12*2C] fp32
        st_all.record_stream(main)          # allocated on the side stream, consumed by the norms on the main stream
        state.record_stream(side)
        # image conditioning (shared conv + the 12 gamma/beta heads as one grouped conv): independent of the encoder below, so
        # it runs on its own side stream, concurrently with it (both forked from the main stream: no nested forks)
        hq, wq = H >> self.n_down, W >> self.n_down
        cs = self._cond_stream() if (COND_SIDE and not ops.SERIALIZE) else main
        if cs is not main:
            stepgraph.fork(cs, main)
            img.record_stream(cs)
        with torch.cuda.stream(cs):
            seg = ops.resize_nearest(img, hq, wq)
            actv = L["shared"].fwd(seg, act=ACT_RELU)                   # [N,h,w,12*nh]
            gm = GB_GROUP_MAJOR and actv.dtype == torch.bfloat16 and nh % 64 == 0 and (2 * C) % 64 == 0
            gb_events, gpc = [], 2 * self.n_blocks
            if gm:
                gb_all = torch.empty((2 * self.n_blocks, N, hq, wq, 2 * C), dtype=actv.dtype, device=actv.device)      # [12,N,h,w,2C]
                nch = GB_FWD_CHUNKS if (cs is not main and GB_FWD_CHUNKS > 1 and (2 * self.n_blocks) % GB_FWD_CHUNKS == 0) else 1
                gpc = 2 * self.n_blocks // nch                         # groups per chunk
                for j in range(nch):
                    L["gb"].fwd_group_major(actv, gb_all, groups=None if nch == 1 else (j * gpc, (j + 1) * gpc))
                    if nch > 1:
                        ev = torch.cuda.Event()
                        ev.record(cs)
                        gb_events.append(ev)
            else:
                gb_all = L["gb"].fwd(actv)                              # [N,h,w,12*2C]
        if cs is not main:
            for t in (seg, actv, gb_all):
                t.record_stream(main)
        # encoder
        enc = []
        x = L["stem"].fwd(img)
        a, s = ops.in_norm_fwd(x, self.ngf, act=ACT_RELU)
        enc.append((img, x, s, a))
        c = self.ngf
        for i in range(self.n_down):
            xin = a
            x = L[f"down{i}"].fwd(xin)
            c *= 2
            a, s = ops.in_norm_fwd(x, c, act=ACT_RELU)
            enc.append((xin, x, s, a))
        # MAT residual blocks
        main.wait_stream(side)                                          # st_all is needed from here on
        if cs is not main and not gb_events:
            main.wait_stream(cs)                                        # ... and the gamma/beta maps
        blocks = []
        x = a
        nA = sA = None
        gb_waited = -1
        for b in range(self.n_blocks):
            if gb_events:                                               # chunked producer: block b reads the planes 2 b .. 2 b + 2
                need = min(2 * b + 2, 2 * self.n_blocks - 1) // gpc
                while gb_waited < need:
                    gb_waited += 1
                    main.wait_event(gb_events[gb_waited])
            o0, o1 = (2 * b) * 2 * C, (2 * b + 1) * 2 * C
            G0, G1 = self._gb_plane(gb_all, 2 * b), self._gb_plane(gb_all, 2 * b + 1)
            if FUSE_NORM_FWD:
                # every MAT norm but the first is applied in the epilogue of the conv that produces its input
                # (s2p_conv2d_fwd_mat): conv_0 -> norm_1, and conv_1 + skip -> norm_0 of the NEXT block
                if nA is None:
                    nA, sA = ops.in_norm_fwd(x, C, *G0, st_all, o0, ACT_LRELU, LRELU)
                # (conv_0's own output is only the backward's input: a forward that saves nothing does not write it)
                c0, nB, sB = L[f"b{b}c0"].fwd_mat(nA, *G1, st_all, o1, ACT_LRELU, LRELU, want_y=save)
                blocks.append((x, sA, nA, c0, sB, nB))
                if b + 1 < self.n_blocks:
                    o0n = (2 * b + 2) * 2 * C
                    x, nA, sA = L[f"b{b}c1"].fwd_mat(nB, *self._gb_plane(gb_all, 2 * b + 2), st_all, o0n, ACT_LRELU, LRELU, aux=x, epi=EPI_ADD)
                else:
                    x = L[f"b{b}c1"].fwd(nB, aux=x, epi=EPI_ADD)
                continue
            nA, sA = ops.in_norm_fwd(x, C, *G0, st_all, o0, ACT_LRELU, LRELU)
            c0 = L[f"b{b}c0"].fwd(nA)
            nB, sB = ops.in_norm_fwd(c0, C, *G1, st_all, o1, ACT_LRELU, LRELU)
            xn = L[f"b{b}c1"].fwd(nB, aux=x, epi=EPI_ADD)
            blocks.append((x, sA, nA, c0, sB, nB))
            x = xn
        if gb_events:
            main.wait_stream(cs)                                        # (the conditioning stream re-joins: a captured segment may end)
        # decoder
        dec = []
        for i in range(self.n_down):
            xin = x
            u = L[f"up{i}"].fwd(xin)
            c //= 2
            x, s = ops.in_norm_fwd(u, c, act=ACT_RELU)
            dec.append((xin, u, s, x))
        out = L["out"].fwd(x, act=ACT_TANH)
        if save:
            ctx.update(hs=hs, st_all=st_all, seg=seg, actv=actv, gb_all=gb_all, enc=enc, blocks=blocks, dec=dec,
                       out=out, last=x)
        return out, ctx

    def _gb_plane(self, t, k):
        """(tensor, channel offset) of norm k's gamma|beta plane (or its gradient) in either layout of `t`."""
        return (t[k], 0) if t.dim() == 5 else (t, k * 2 * self.c_mid)

    def _cond_stream(self):
        s = getattr(self, "_cside", None)
        if s is None:
            s = self._cside = torch.cuda.Stream()
        return s

    def _wgrad_stream(self):
        s = getattr(self, "_wside", None)
        if s is None:
            s = self._wside = torch.cuda.Stream()
        return s

    def _side_stream(self):
        if ops.SERIALIZE:
            return torch.cuda.current_stream()
        s = getattr(self, "_side", None)
        if s is None:
            s = self._side = torch.cuda.Stream()
        return s

    @property
    def L_oct(self):
        return self.posenc_L

    def bwd_nhwc(self, ctx, d_out):
        """Backward from d(loss)/d(out) (NHWC).  All parameter gradients are accumulated into the flat grad buffer."""
        L, C, nh = self.lay, self.c_mid, self.nhidden
        out = ctx["out"]
        main = torch.cuda.current_stream()
        ws0 = self._wgrad_stream() if (DEC_WGRAD_SIDE and WGRAD_SIDE and not ops.SERIALIZE) else None

        def dec_wgrad(layer, x, dy):
            """Decoder weight gradients on the weight-gradient stream: the decoder's backward is a serial chain of short launches
            (dgrad -> norm backward -> dgrad ...) and these do not feed it."""
            if ws0 is None:
                layer.wgrad(x, dy)
                return
            stepgraph.fork(ws0, main)
            x.record_stream(ws0); dy.record_stream(ws0)
            with torch.cuda.stream(ws0):
                layer.wgrad(x, dy)

        dpre = ops.act_bwd(d_out, out, ACT_TANH)
        dec_wgrad(L["out"], ctx["last"], dpre)
        dx = L["out"].dgrad(dpre, ctx["last"].shape)
        c = self.ngf
        for i in reversed(range(self.n_down)):
            xin, u, s, a = ctx["dec"][i]
            cc = u.shape[3]
            du = ops.in_bwd(dx, u, cc, s, act=ACT_RELU)
            dec_wgrad(L[f"up{i}"], xin, du)
            dx = L[f"up{i}"].dgrad(du, xin.shape)
        gb_all, st_all = ctx["gb_all"], ctx["st_all"]
        dgb_all = torch.empty_like(gb_all)
        dst_all = torch.empty_like(st_all)                              # [N, 12*2C]: filled by the 12 MAT backward passes
        N = st_all.shape[0]
        # The 12 block convs' weight gradients do not feed the backward chain: they are deferred and run as ONE batched
        # launch behind it (12 x 16 output tiles x K-splits fill the chip without split-K 16 atomics); their dY tensors
        # (12 x 14 MB at bs 64) simply stay alive until then.
        # With WGRAD_SIDE they go to a side stream in batches of WGRAD_CHUNK_BLOCKS blocks while the chain is still running:
        # the chain alternates MFMA-bound convs (442 workgroups on 512 slots) with HBM-bound MAT backward passes, and the
        # weight-gradient workgroups take the MFMA time those leave idle.
        ws = self._wgrad_stream() if (WGRAD_SIDE and not ops.SERIALIZE) else None

        def side_wgrads(jobs, extra=None):
            if ws is None:
                ConvLayer.wgrad_many(jobs)
                if extra is not None:
                    extra()
                return
            stepgraph.fork(ws, main)
            for _, x, dy in jobs:
                x.record_stream(ws); dy.record_stream(ws)
            with torch.cuda.stream(ws):
                ConvLayer.wgrad_many(jobs)
                if extra is not None:
                    extra()

        actv, seg = ctx["actv"], ctx["seg"]
        if ws is not None:
            actv.record_stream(ws); dgb_all.record_stream(ws)
        hook = self.on_early_grads
        chunked = GB_WGRAD_CHUNKED and hook is not None and ws is not None and WGRAD_CHUNK_BLOCKS > 0 and gb_all.dim() == 5
        if getattr(self, "_n_buckets", None) is None:
            self._n_buckets = len(self.early_buckets())
        n_buckets = self._n_buckets if (hook is not None and chunked) else 1
        pending = []              # batches issued on the side stream whose bucket has not been handed to the exchange yet

        def gb_wgrad(b_lo, b_hi):             # the gamma/beta heads of the norms of blocks b_lo..b_hi (groups 2 b_lo .. 2 b_hi + 1)
            return lambda: L["gb"].wgrad(actv, dgb_all, groups=(2 * b_lo, 2 * b_hi + 2))

        def hand_over():
            """The previous batch ran under the blocks just finished: re-join the side stream and start that bucket's exchange."""
            nonlocal ws
            if hook is not None and chunked and pending:
                if stepgraph.capturing_now():
                    main.wait_stream(ws)                    # a graph segment ends at the hook: the side stream has to re-join first
                    hook(pending.pop(0))
                else:
                    # eager: only the COMMUNICATION stream waits for the batch (an event on the side stream); the dgrad chain of the
                    # next blocks on the main stream does not stall behind the weight gradients that were meant to run under it
                    ev = torch.cuda.Event()
                    ev.record(ws)
                    hook(pending.pop(0), after=ev)

        wjobs = []
        for k, b in enumerate(reversed(range(self.n_blocks))):
            x, sA, nA, c0, sB, nB = ctx["blocks"][b]
            o0, o1 = (2 * b) * 2 * C, (2 * b + 1) * 2 * C
            G0, G1 = self._gb_plane(gb_all, 2 * b), self._gb_plane(gb_all, 2 * b + 1)
            D0, D1 = self._gb_plane(dgb_all, 2 * b), self._gb_plane(dgb_all, 2 * b + 1)
            if FUSE_NORM_BWD:
                # each MAT norm's backward runs in the epilogue of the dgrad of the conv it fed (s2p_conv2d_dgrad_mat):
                # dL/d(norm output) never goes to HBM; the skip gradient is added in the second launch
                wjobs.append((L[f"b{b}c1"], nB, dx))
                d_c0 = L[f"b{b}c1"].dgrad_mat(dx, c0, sB, *G1, st_all, o1, ACT_LRELU, LRELU, *D1, dst_all, o1)
                wjobs.append((L[f"b{b}c0"], nA, d_c0))
                dx = L[f"b{b}c0"].dgrad_mat(d_c0, x, sA, *G0, st_all, o0, ACT_LRELU, LRELU, *D0, dst_all, o0, res=dx)
            else:
                wjobs.append((L[f"b{b}c1"], nB, dx))
                d_nB = L[f"b{b}c1"].dgrad(dx, nB.shape)
                d_c0 = ops.in_bwd(d_nB, c0, C, sB, *G1, st_all, o1, ACT_LRELU, LRELU, *D1, dst_all, o1)
                wjobs.append((L[f"b{b}c0"], nA, d_c0))
                d_nA = L[f"b{b}c0"].dgrad(d_c0, nA.shape)
                if FUSE_SKIP_ADD:       # the skip-connection gradient dx is added in the same launch: no separate add pass
                    dx = ops.in_bwd(d_nA, x, C, sA, *G0, st_all, o0, ACT_LRELU, LRELU, *D0, dst_all, o0, res=dx)
                else:
                    d_xb = ops.in_bwd(d_nA, x, C, sA, *G0, st_all, o0, ACT_LRELU, LRELU, *D0, dst_all, o0)
                    dx = ops.add(dx, d_xb, out=d_xb)
            if ws is not None and WGRAD_CHUNK_BLOCKS > 0 and (k + 1) % WGRAD_CHUNK_BLOCKS == 0 and k + 1 < self.n_blocks:
                hand_over()
                side_wgrads(wjobs, gb_wgrad(b, b + WGRAD_CHUNK_BLOCKS - 1) if chunked else None)
                pending.append((k + 1) // WGRAD_CHUNK_BLOCKS - 1)       # this batch completes bucket 0, 1, ...
                wjobs = []
        # the rest of the block convs' weight gradients + the image-conditioning branch's (batched)
        hand_over()
        if chunked:
            rest = len(wjobs) // 2                                       # blocks still in wjobs: 0 .. rest - 1
            side_wgrads(wjobs, gb_wgrad(0, rest - 1) if rest > 0 else None)
        else:
            side_wgrads(wjobs, lambda: L["gb"].wgrad(actv, dgb_all))
        if self.on_tail_final is not None:
            # one rank: this step's Adam update of the tail (87 % of the parameters) runs here, behind the last weight gradients on their
            # stream and under the rest of the backward, instead of alone on the chip after it (the optimizer then finishes the head)
            if ws is not None:
                with torch.cuda.stream(ws):
                    self.on_tail_final()
            else:
                self.on_tail_final()
        # every gradient of the flat buffer's tail [early_grad_offset, end) is final once these are done (data-parallel hook:
        # the trainer starts the last bucket's all-reduce here, under the rest of this backward; a graph segment ends at the
        # hook, so the side stream re-joins first)
        if hook is not None:
            if ws is not None:
                main.wait_stream(ws)
                ws = None
            hook(n_buckets - 1 if chunked else None)
        # backward of the image-conditioning branch on its side stream, concurrent with the encoder backward below
        cs = self._cond_stream() if (COND_SIDE and not ops.SERIALIZE) else main
        if cs is not main:
            stepgraph.fork(cs, main)
            dgb_all.record_stream(cs); actv.record_stream(cs)
        with torch.cuda.stream(cs):
            d_actv = L["gb"].dgrad(dgb_all, actv.shape, aux=actv, epi=EPI_MUL_ACTGRAD, aux_act=ACT_RELU)
            L["shared"].wgrad(seg, d_actv)
        # state path backward on the side stream, overlapped with the encoder backward below
        side = self._side_stream() if STATE_SIDE_BWD else main
        stepgraph.fork(side, main)
        dst_all.record_stream(side)
        with torch.cuda.stream(side):
            hs = ctx["hs"]
            pk = L["fc_state"].pk
            dh = ops.linear_bwd(hs[-1], dst_all, None, pk.w_bwd, pk.Cpad, pk.C, pk.R, ACT_NONE, 0.0, pk.gw, pk.gb)
            for i in reversed(range(self.n_mlp)):
                pk = L[f"fc{i}"].pk              # LeakyReLU derivative (from the saved output hs[i+1]) folded into the operands
                dh = ops.linear_bwd(hs[i], dh, hs[i + 1], pk.w_bwd, pk.Cpad, pk.C, pk.R, ACT_LRELU, LRELU, pk.gw, pk.gb,
                                    need_dx=i > 0)
        # encoder
        for i in reversed(range(self.n_down)):
            xin, x, s, a = ctx["enc"][i + 1]
            dxe = ops.in_bwd(dx, x, x.shape[3], s, act=ACT_RELU)
            L[f"down{i}"].wgrad(xin, dxe)
            dx = L[f"down{i}"].dgrad(dxe, xin.shape)
        img, x, s, a = ctx["enc"][0]
        dxe = ops.in_bwd(dx, x, self.ngf, s, act=ACT_RELU)
        L["stem"].wgrad(img, dxe)
        main.wait_stream(side)
        if cs is not main:
            main.wait_stream(cs)
        if ws is not None:
            main.wait_stream(ws)

    # ---- public torch-style API --------------------------------------------------------------------------------
    def forward(self, prev_image, state):
        """prev_image: fp32 NCHW [N,3,H,W] in [-1,1]; state: fp32 [N,S] -> fp32 NCHW [N,3,H,W] (autograd-aware)."""
        self._require_ready()
        from ..autograd_nodes import generator_apply, nhwc_to_nchw_apply
        return nhwc_to_nchw_apply(generator_apply(self, prev_image, state), 3)

    def forward_nhwc(self, prev_image, state):
        self._require_ready()
        from ..autograd_nodes import generator_apply
        return generator_apply(self, prev_image, state)

"""Thin layer helpers over the raw HIP ops with hand-written backward (no autograd inside the networks:
each network is ONE coarse autograd node, which is what lets gradients go straight into the flat grad buffer)."""
import torch
import torch.nn as nn

from ... import ops
from ..._lib import ACT_LRELU, ACT_NONE, ACT_RELU, ACT_TANH, EPI_ADD, EPI_MUL_ACTGRAD, EPI_STORE, chunk_elems
from ...ops import ConvGeom, pad_to


class ConvLayer:
    """One (possibly fused / grouped) conv launch bound to a Packed operand."""

    def __init__(self, packed, geom, cin_real=None, cout_real=None):
        self.pk, self.geom = packed, geom
        self.cin_real = cin_real if cin_real is not None else geom.cin
        self.cout_real = cout_real if cout_real is not None else geom.cout

    def cin_pad(self, dtype):
        return pad_to(self.geom.cin, chunk_elems(dtype))

    def fwd(self, x, act=ACT_NONE, slope=0.2, aux=None, epi=EPI_STORE, y_pitch=None):
        return ops.conv_fwd(self.geom, x, self.pk.w_fwd, self.pk.bias, self.cin_pad(x.dtype), y_pitch=y_pitch,
                            act=act, slope=slope, aux=aux, epi=epi)

    def _group_major(self, t5):
        """This grouped conv's geometry with the PRODUCED side group-major: t5 is [groups, N, H, W, pitch] (one whole NHWC
        tensor per group) instead of channel slices of one wide tensor."""
        g = self.geom
        assert t5.dim() == 5 and t5.shape[0] == g.groups and t5.is_contiguous()
        key = t5.stride(0)
        gm = self.__dict__.setdefault("_gm", {})
        if key not in gm:
            gm[key] = ConvGeom(g.cin, g.cout, g.k, g.stride, g.pad, g.transposed, g.reflect, g.groups, g.output_padding,
                               x_gstride=g.x_gstride, y_gstride=key, net=g.net)
        return gm[key]

    def fwd_group_major(self, x, out5, groups=None):
        """Grouped conv whose group g writes the whole tensor out5[g] ([groups, N, Ho, Wo, pitch]).  groups=(g0, g1): only the groups
        g0 .. g1 - 1 (the generator produces the gamma|beta planes in chunks that run ahead of the ResBlk chain)."""
        gm = self._group_major(out5)
        if groups is None:
            ops.conv_fwd(gm, x, self.pk.w_fwd, self.pk.bias, self.cin_pad(x.dtype), y_pitch=out5.shape[4], out=out5[0])
            return out5
        g0, g1 = groups
        sub = self.__dict__.setdefault("_gm_sub", {})
        key = (out5.stride(0), g1 - g0)
        if key not in sub:
            sub[key] = ConvGeom(gm.cin, gm.cout, gm.k, gm.stride, gm.pad, gm.transposed, gm.reflect, g1 - g0, gm.output_padding,
                                x_gstride=gm.x_gstride, y_gstride=gm.y_gstride, net=gm.net)
        bias = self.pk.bias[g0 * gm.cout:g1 * gm.cout] if self.pk.bias is not None else None
        ops.conv_fwd(sub[key], x, self.pk.w_fwd[g0:g1], bias, self.cin_pad(x.dtype), y_pitch=out5.shape[4], out=out5[g0],
                     x_off=g0 * gm.x_gstride)
        return out5

    def fwd_mat(self, x, gb, gb_off, gb_st, st_off, act, slope, aux=None, epi=EPI_STORE, want_y=True):
        """conv (+ residual) -> InstanceNorm -> MAT modulation -> activation; returns (conv output, modulated, stats).
        want_y=False: the conv output is only kept for a backward pass -- a fused launch then does not write it (returns None)."""
        return ops.conv_fwd_mat(self.geom, x, self.pk.w_fwd, self.pk.bias, self.cin_pad(x.dtype), gb, gb_off, gb_st, st_off,
                                act=act, slope=slope, aux=aux, epi=epi, want_y=want_y)

    def dgrad(self, dy, x_shape, aux=None, epi=EPI_STORE, aux_act=ACT_NONE, slope=0.2, aux2=None):
        if dy.dim() == 5:           # group-major dy ([groups, N, Ho, Wo, pitch]: see fwd_group_major)
            return ops.conv_dgrad(self._group_major(dy), dy[0], self.pk.w_bwd, tuple(x_shape), self.cin_pad(dy.dtype), aux=aux,
                                  epi=epi, aux_act=aux_act, slope=slope, aux2=aux2)
        return ops.conv_dgrad(self.geom, dy, self.pk.w_bwd, tuple(x_shape), self.cin_pad(dy.dtype), aux=aux, epi=epi,
                              aux_act=aux_act, slope=slope, aux2=aux2)

    def dgrad_mat(self, dy, xn, stats, gb, gb_off, gb_st, st_off, act, slope, dgb, dgb_off, dgb_st, dst_off, res=None, aux=None):
        """dgrad of this conv followed by the backward of the MAT norm that produced its input (one launch where the
        plane-resident kernel applies): returns dL/d(norm input) (+ res)."""
        return ops.conv_dgrad_mat(self.geom, dy, self.pk.w_bwd, xn, self.cin_pad(dy.dtype), stats, gb, gb_off, gb_st, st_off,
                                  act, slope, dgb, dgb_off, dgb_st, dst_off, res=res, aux=aux)

    def wgrad(self, x, dy, groups=None):
        """Accumulate weight (and bias) gradients straight into the flat grad buffer.  groups=(g0, g1): only the groups g0 .. g1 - 1
        of a grouped conv (the generator's gamma/beta heads in batches that follow the ResBlk backward)."""
        g = self.geom
        if g.groups > 1 and x.dtype == torch.bfloat16 and g.k == 3 and g.stride == 1 and g.pad == 1 and not g.transposed \
                and g.groups <= 16 and g.cin % 64 == 0 and g.cout % 64 == 0:
            # grouped 3x3 conv (the 12 gamma/beta heads): one job per group of the batched slab kernel
            one = ConvGeom(g.cin, g.cout, 3, 1, 1, net=g.net)
            gw = self.pk.gw.view(g.groups, -1)
            gb = self.pk.gb.view(g.groups, -1) if self.pk.gb is not None else None
            sel = range(g.groups) if groups is None else range(groups[0], groups[1])
            if dy.dim() == 5:       # group-major dy: one whole tensor per group
                jobs = [(x, i * g.x_gstride, dy[i], 0, gw[i], gb[i] if gb is not None else None) for i in sel]
            else:
                jobs = [(x, i * g.x_gstride, dy, i * g.y_gstride, gw[i], gb[i] if gb is not None else None) for i in sel]
            ops.conv_wgrad_batched(one, jobs, g.cin, self.cin_real, self.cout_real)
            return
        if groups is not None:
            raise RuntimeError("a group range is only wired for the batched slab weight-gradient path")
        if dy.dim() == 5:
            raise RuntimeError("group-major dY is only wired for the batched slab weight-gradient path (bf16, 3x3, 64-channel multiples)")
        if g.groups == 1 and g.cout == 1 and x.dtype == torch.bfloat16 and g.stride == 1 and not g.transposed and not g.reflect:
            # PatchGAN logit head: activation-stationary kernel behind the batched entry point (it needs a workspace)
            ops.conv_wgrad_batched(g, [(x, 0, dy, 0, self.pk.gw, self.pk.gb)], self.cin_pad(x.dtype), self.cin_real,
                                   self.cout_real)
            return
        ops.conv_wgrad(self.geom, x, dy, self.pk.gw, self.cin_pad(x.dtype), self.cin_real, self.cout_real,
                       dw_gstride=self.pk.gw_gstride, db=self.pk.gb)      # bias gradient fused into the wgrad pass

    @staticmethod
    def wgrad_many(items):
        """items: list of (layer, x, dy) of layers with the SAME geometry: one batched launch when the slab kernel applies
        (s2p_conv2d_wgrad_batched falls back to one launch per job otherwise)."""
        if not items:
            return
        l0, x0, _ = items[0]
        g = l0.geom
        same = all(l.geom.__dict__ == g.__dict__ and x.shape == x0.shape and l.cin_real == l0.cin_real
                   and l.cout_real == l0.cout_real for l, x, _ in items)
        if not same or g.groups != 1 or len(items) > 16:
            for l, x, dy in items:
                l.wgrad(x, dy)
            return
        jobs = [(x, 0, dy, 0, l.pk.gw, l.pk.gb) for l, x, dy in items]
        ops.conv_wgrad_batched(g, jobs, l0.cin_pad(x0.dtype), l0.cin_real, l0.cout_real)


def make_conv_param(cout, cin, k):
    return nn.Parameter(torch.zeros(cout, cin, k, k))

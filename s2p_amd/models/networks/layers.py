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

    def dgrad(self, dy, x_shape, aux=None, epi=EPI_STORE, aux_act=ACT_NONE, slope=0.2, aux2=None):
        return ops.conv_dgrad(self.geom, dy, self.pk.w_bwd, tuple(x_shape), self.cin_pad(dy.dtype), aux=aux, epi=epi,
                              aux_act=aux_act, slope=slope, aux2=aux2)

    def wgrad(self, x, dy):
        """Accumulate weight (and bias) gradients straight into the flat grad buffer."""
        ops.conv_wgrad(self.geom, x, dy, self.pk.gw, self.cin_pad(x.dtype), self.cin_real, self.cout_real,
                       dw_gstride=self.pk.gw_gstride, db=self.pk.gb)      # bias gradient fused into the wgrad pass


def make_conv_param(cout, cin, k):
    return nn.Parameter(torch.zeros(cout, cin, k, k))

"""Losses of the S2P train step: hinge GAN, discriminator feature matching, VGG19 perceptual, pixel L1
(rebuttal.md:71,135,187-188 name L1 + GAN + perceptual with an ImageNet VGG; weights/structure per SPEC.md).

VGG19 runs as explicit HIP launches (conv3x3+ReLU fused epilogue, 2x2 max-pool); its backward is dgrad-only
(frozen weights), on the fake half of the batch only, with the ReLU masks and the perceptual-loss taps folded into
the dgrad epilogue.  ImageNet weights cannot be fetched offline: `--vgg_weights <file>` loads a torchvision-style
state_dict; otherwise seeded He-normal stand-in weights are used (throughput is identical; documented in DESIGN.md).
"""
import torch
import torch.nn as nn

from ... import ops
from ..._lib import ACT_RELU, EPI_MUL_ACTGRAD, chunk_elems
from ...ops import ConvGeom
from .base_network import BaseNetwork
from .generator import _Conv
from .layers import ConvLayer

VGG_CFG = [("conv1_1", 3, 64), ("conv1_2", 64, 64), "P",
           ("conv2_1", 64, 128), ("conv2_2", 128, 128), "P",
           ("conv3_1", 128, 256), ("conv3_2", 256, 256), ("conv3_3", 256, 256), ("conv3_4", 256, 256), "P",
           ("conv4_1", 256, 512), ("conv4_2", 512, 512), ("conv4_3", 512, 512), ("conv4_4", 512, 512), "P",
           ("conv5_1", 512, 512)]
VGG_TAPS = ("conv1_1", "conv2_1", "conv3_1", "conv4_1", "conv5_1")
VGG_WEIGHTS = (1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0)
# torchvision vgg19.features indices of the convs above (for loading a torchvision state_dict)
TORCHVISION_IDX = dict(conv1_1=0, conv1_2=2, conv2_1=5, conv2_2=7, conv3_1=10, conv3_2=12, conv3_3=14, conv3_4=16,
                       conv4_1=19, conv4_2=21, conv4_3=23, conv4_4=25, conv5_1=28)


class VGG19(BaseNetwork):
    def __init__(self, opt=None):
        super().__init__()
        for item in VGG_CFG:
            if item != "P":
                name, cin, cout = item
                setattr(self, name, _Conv(cin, cout, 3, bias=True))
        for p in self.parameters():
            p.requires_grad_(False)

    def init_standin(self, seed=1234):
        """Seeded He-normal stand-in weights (ImageNet weights are not obtainable offline)."""
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for name, p in self.named_parameters():
                if name.endswith(".bias"):
                    p.zero_()
                else:
                    fan_in = p.shape[1] * p.shape[2] * p.shape[3]
                    p.copy_(torch.randn(p.shape, generator=g) * (2.0 / fan_in) ** 0.5)
        if self.finalized:
            self.store.repack()

    def load_torchvision(self, path):
        sd = torch.load(path, map_location="cpu")
        mine = {}
        for name, idx in TORCHVISION_IDX.items():
            for suffix in ("weight", "bias"):
                for key in (f"features.{idx}.{suffix}", f"{idx}.{suffix}", f"{name}.{suffix}"):
                    if key in sd:
                        mine[f"{name}.{suffix}"] = sd[key]
                        break
        self.load_state_dict(mine, strict=True)

    def _declare_packs(self, dt):
        self.lay = {}
        for item in VGG_CFG:
            if item == "P":
                continue
            name, cin, cout = item
            m = getattr(self, name)
            self.store.add(m.weight, "conv"); self.store.add(m.bias, "bias")
            pk = self.store.pack(name, [m.weight], [m.bias], dtype=dt)
            self.lay[name] = ConvLayer(pk, ConvGeom(cin, cout, 3, 1, 1, net="VGG"))

    def fwd_nhwc(self, x):
        """x: NHWC [B,H,W,ce] (3 real channels).  Returns (tap features, ctx)."""
        self._require_ready()
        acts = []       # (kind, name, input, output)
        h = x
        taps = []
        for item in VGG_CFG:
            if item == "P":
                o = ops.maxpool_fwd(h)
                acts.append(("P", None, h, o))
            else:
                name = item[0]
                o = self.lay[name].fwd(h, act=ACT_RELU)
                acts.append(("C", name, h, o))
                if name in VGG_TAPS:
                    taps.append(o)
            h = o
        return taps, acts

    def bwd_nhwc(self, acts, tap_grads, n_fake):
        """Backward through the first `n_fake` samples only.  tap_grads[k]: d(loss)/d(tap k) for those samples
        (un-masked).  Returns d(loss)/dx [n_fake,H,W,ce]."""
        tapg = dict(zip(VGG_TAPS, tap_grads))
        d = None            # gradient w.r.t. the PRE-activation of the conv whose output we stand at
        for kind, name, xin, out in reversed(acts):
            xin_f, out_f = xin[:n_fake], out[:n_fake]
            if kind == "P":
                # d is dpre of the conv after the pool -> already converted below; here we hold d(pool out)
                d = ops.maxpool_bwd(d, xin_f)          # = dpre of the conv that produced xin (ReLU mask fused)
                continue
            # here `d` must be dpre of THIS conv.  For the last conv it comes from its tap only.
            if d is None:
                d = ops.act_bwd(tapg[name], out_f, ACT_RELU)
            lay = self.lay[name]
            if name == "conv1_1":
                return lay.dgrad(d, xin_f.shape)
            # gradient w.r.t. this conv's input; the input is either a ReLU output (fold mask + optional tap) or a pool
            prev_is_pool = self._prev_kind(acts, name) == "P"
            if prev_is_pool:
                d = lay.dgrad(d, xin_f.shape)          # d(pool out); masked by the pool backward next
            else:
                prev_name = self._prev_name(acts, name)
                d = lay.dgrad(d, xin_f.shape, aux=xin_f, epi=EPI_MUL_ACTGRAD, aux_act=ACT_RELU,
                              aux2=tapg.get(prev_name))
        raise RuntimeError("unreachable")

    @staticmethod
    def _prev_kind(acts, name):
        for i, (kind, n, _, _) in enumerate(acts):
            if n == name:
                return acts[i - 1][0] if i > 0 else None
        return None

    @staticmethod
    def _prev_name(acts, name):
        for i, (kind, n, _, _) in enumerate(acts):
            if n == name:
                return acts[i - 1][1] if i > 0 else None
        return None

"""SLAC encoder / decoder conv stacks on HIP: the first consumer of the generated frames
(`rlkit/torch/slac/network/latent.py:116-171` Encoder, `:55-113` Decoder, image_size 100 branches).  Mirrors the
reference call surface -- `Encoder(input_dim, output_dim, image_size)(x [B,S,C,H,W]) -> [B,S,256]`,
`Decoder(input_dim, output_dim, std, image_size)(z [B,S,L]) -> (mean [B,S,3,100,100], std)` -- and loads the reference
state_dict (`net.<2i>.weight/bias`).  Every layer is one implicit-GEMM launch with bias + LeakyReLU(0.2) fused in the
epilogue (conv-transpose layers run as sub-pixel phases); activations stay NHWC in HBM between layers.
Forward (inference) only this round; the kernels' dgrad / wgrad entry points already cover the backward.
"""
import torch

from . import ops
from ._lib import ACT_LRELU, chunk_elems
from .ops import ConvGeom, pad_to

ENCODER_100 = [("conv", 3, 32, 5, 2, 2, 0), ("conv", 32, 64, 3, 2, 1, 0), ("conv", 64, 128, 3, 2, 1, 0),
               ("conv", 128, 256, 3, 2, 1, 0), ("conv", 256, 256, 3, 2, 1, 0), ("conv", 256, 256, 4, 1, 0, 0)]
DECODER_100 = [("convT", 288, 256, 4, 1, 0, 0), ("convT", 256, 256, 3, 2, 1, 0), ("convT", 256, 128, 3, 2, 1, 0),
               ("convT", 128, 64, 3, 2, 1, 0), ("convT", 64, 32, 3, 2, 1, 1), ("convT", 32, 3, 5, 2, 2, 1)]


class _Stack:
    def __init__(self, spec, dtype, device):
        self.spec, self.dtype, self.device = spec, dtype, torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the SLAC conv stacks run on a HIP device only (no CPU fallback)")
        self.layers = None

    def load_state_dict(self, sd):
        ce = chunk_elems(self.dtype)
        self.layers = []
        for i, (kind, cin, cout, k, s, pad, op) in enumerate(self.spec):
            w = torch.as_tensor(sd[f"net.{2 * i}.weight"], dtype=torch.float32)
            b = torch.as_tensor(sd[f"net.{2 * i}.bias"], dtype=torch.float32)
            cin_pad = pad_to(cin, ce)
            wp = torch.zeros(cout, k * k, cin_pad)
            if kind == "conv":                                   # [Cout,Cin,kh,kw] -> [Cout][tap][Cin]
                wp[:, :, :cin] = w.permute(0, 2, 3, 1).reshape(cout, k * k, cin)
            else:                                                # ConvTranspose2d [Cin,Cout,kh,kw] -> [Cout][tap][Cin]
                wp[:, :, :cin] = w.permute(1, 2, 3, 0).reshape(cout, k * k, cin)
            geom = ConvGeom(cin, cout, k, s, pad, transposed=(kind == "convT"), output_padding=op)
            self.layers.append((geom, wp.to(self.dtype).to(self.device).contiguous(), b.to(self.device).contiguous(), cin_pad))
        return self

    @torch.no_grad()
    def run(self, x_nhwc):
        h = x_nhwc
        for geom, w, b, cin_pad in self.layers:
            h = ops.conv_fwd(geom, h, w, b, cin_pad, act=ACT_LRELU, slope=0.2)
        return h


class Encoder(_Stack):
    def __init__(self, input_dim=3, output_dim=256, image_size=100, dtype=torch.float32, device="cuda:0"):
        if image_size != 100 or input_dim != 3 or output_dim != 256:
            raise NotImplementedError("only the image_size=100 configuration used by the shipped run scripts is built")
        super().__init__(ENCODER_100, dtype, device)

    def __call__(self, x):
        """x: fp32 [B,S,3,100,100] in [0,1] (or uint8 NHWC frames [B,S,100,100,3]) -> fp32 [B,S,256]."""
        if x.dtype == torch.uint8:
            B, S = x.shape[:2]
            frames = x.reshape(B * S, *x.shape[2:]).to(self.device).contiguous()
            h = ops.u8_to_nhwc(frames, self.dtype, chunk_elems(self.dtype))      # [-1,1]; SLAC wants [0,1]:
            h = (h.float() + 1.0).mul_(0.5).to(self.dtype)
            h[..., 3:] = 0
        else:
            B, S, C, H, W = x.shape
            h = ops.nchw_to_nhwc(x.reshape(B * S, C, H, W).to(self.device, torch.float32).contiguous(), self.dtype,
                                 chunk_elems(self.dtype))
        y = self.run(h)                                            # [B*S,1,1,256]
        return y.reshape(B, S, -1).float()


class Decoder(_Stack):
    def __init__(self, input_dim=288, output_dim=3, std=1.0, image_size=100, dtype=torch.float32, device="cuda:0"):
        if image_size != 100 or input_dim != 288 or output_dim != 3:
            raise NotImplementedError("only the image_size=100 configuration used by the shipped run scripts is built")
        super().__init__(DECODER_100, dtype, device)
        self.std = std

    def __call__(self, z):
        """z: fp32 [B,S,288] -> (mean fp32 [B,S,3,100,100], std tensor filled with self.std)."""
        B, S, L = z.shape
        h = z.reshape(B * S, 1, 1, L).to(self.device, self.dtype).contiguous()
        y = self.run(h)                                            # [B*S,100,100,pitch]
        img = ops.nhwc_to_nchw(y, 3).reshape(B, S, 3, y.shape[1], y.shape[2])
        return img, torch.full_like(img, self.std)

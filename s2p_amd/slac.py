"""SLAC encoder / decoder conv stacks on HIP: the first consumer of the generated frames
(`rlkit/torch/slac/network/latent.py:116-171` Encoder, `:55-113` Decoder, image_size 100 branches), trained on every RL
step (`iql_trainer.py:348-350`).  Mirrors the reference call surface -- `Encoder(input_dim, output_dim, image_size)(x
[B,S,C,H,W]) -> [B,S,256]`, `Decoder(input_dim, output_dim, std, image_size)(z [B,S,L]) -> (mean [B,S,3,100,100], std)`
-- as `nn.Module`s whose parameters carry the reference's state_dict keys (`net.<2i>.weight/bias`, reference layouts),
so reference checkpoints load natively and any torch optimizer trains them.

Forward: every layer is one implicit-GEMM launch with bias + LeakyReLU(0.2) fused in the epilogue (conv-transpose layers
run as merged sub-pixel phases); activations stay NHWC in HBM between layers.
Backward: the whole stack is ONE autograd node.  Per layer, in reverse: wgrad (bias gradient fused for the gather
form) straight from the saved NHWC activations, and a dgrad whose epilogue multiplies by the LeakyReLU derivative of
the layer below (EPI_MUL_ACTGRAD), so no separate activation-backward pass runs between layers.
"""
import torch
import torch.nn as nn

from . import ops
from ._lib import ACT_LRELU, EPI_MUL_ACTGRAD, chunk_elems
from .ops import ConvGeom, pad_to

ENCODER_100 = [("conv", 3, 32, 5, 2, 2, 0), ("conv", 32, 64, 3, 2, 1, 0), ("conv", 64, 128, 3, 2, 1, 0),
               ("conv", 128, 256, 3, 2, 1, 0), ("conv", 256, 256, 3, 2, 1, 0), ("conv", 256, 256, 4, 1, 0, 0)]
DECODER_100 = [("convT", 288, 256, 4, 1, 0, 0), ("convT", 256, 256, 3, 2, 1, 0), ("convT", 256, 128, 3, 2, 1, 0),
               ("convT", 128, 64, 3, 2, 1, 0), ("convT", 64, 32, 3, 2, 1, 1), ("convT", 32, 3, 5, 2, 2, 1)]
SLOPE = 0.2


class _Layer(nn.Module):
    """Parameter holder under the reference's key (`net.<2i>`); weight in the torch layout of the reference layer."""

    def __init__(self, kind, cin, cout, k):
        super().__init__()
        shape = (cout, cin, k, k) if kind == "conv" else (cin, cout, k, k)
        self.weight = nn.Parameter(torch.zeros(shape))
        self.bias = nn.Parameter(torch.zeros(cout))


class _StackFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, stack, *params):
        packed = stack.packed()
        acts, h = [x], x
        for (geom, cin_pad, _), (wf, _, b) in zip(stack.meta, packed):
            h = ops.conv_fwd(geom, h, wf, b, cin_pad, act=ACT_LRELU, slope=SLOPE)
            acts.append(h)
        ctx.stack, ctx.packed = stack, packed
        ctx.acts = acts if torch.is_grad_enabled() or any(ctx.needs_input_grad) else None
        return h

    @staticmethod
    def backward(ctx, dy):
        stack, acts, packed = ctx.stack, ctx.acts, ctx.packed
        dev = dy.device
        need_dx = ctx.needs_input_grad[0]
        d = ops.act_bwd(dy.contiguous(), acts[-1], ACT_LRELU, SLOPE)     # gradient wrt the last pre-activation
        grads = [None] * (2 * len(stack.meta))
        for i in reversed(range(len(stack.meta))):
            geom, cin_pad, (kind, cin, cout, k) = stack.meta[i]
            x_i = acts[i]
            tr = kind == "convT"
            rows, cols = (cin, cout) if tr else (cout, cin)
            dw = torch.zeros((rows, k * k, cols), dtype=torch.float32, device=dev)
            db = torch.zeros(cout, dtype=torch.float32, device=dev)
            if tr:                                                       # scatter form: bias gradient is a plain channel sum
                ops.conv_wgrad(geom, x_i, d, dw, cin_pad, cin, cout)
                ops.channel_sum(d, cout, db)
            else:
                ops.conv_wgrad(geom, x_i, d, dw, cin_pad, cin, cout, db=db)
            grads[2 * i] = dw.reshape(rows, k, k, cols).permute(0, 3, 1, 2).contiguous()
            grads[2 * i + 1] = db
            if i > 0:                                                    # x_i = lrelu(pre_{i-1}): fold its derivative in
                d = ops.conv_dgrad(geom, d, packed[i][1], tuple(x_i.shape), cin_pad, aux=x_i, epi=EPI_MUL_ACTGRAD,
                                   aux_act=ACT_LRELU, slope=SLOPE)
            elif need_dx:
                d = ops.conv_dgrad(geom, d, packed[i][1], tuple(x_i.shape), cin_pad)
        ctx.acts = None
        return (d if need_dx else None, None) + tuple(grads)


class _Stack(nn.Module):
    def __init__(self, spec, dtype, device):
        super().__init__()
        self.spec, self.dtype = spec, dtype
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("the SLAC conv stacks run on a HIP device only (no CPU fallback)")
        self.net = nn.Module()
        ce = chunk_elems(dtype)
        self.meta = []
        for i, (kind, cin, cout, k, s, pad, op) in enumerate(spec):
            self.net.add_module(str(2 * i), _Layer(kind, cin, cout, k))
            geom = ConvGeom(cin, cout, k, s, pad, transposed=(kind == "convT"), output_padding=op)
            self.meta.append((geom, pad_to(cin, ce), (kind, cin, cout, k)))
        self.to(dev)
        self._pack_key, self._packed = None, None

    @property
    def device(self):
        return self.net._modules["0"].weight.device

    def load_state_dict(self, sd, strict=True):
        """Accepts the reference's state_dict (tensors or numpy arrays); returns self for chaining."""
        sd = {k: torch.as_tensor(v, dtype=torch.float32) for k, v in sd.items()}
        super().load_state_dict(sd, strict=strict)
        return self

    def layer_params(self):
        out = []
        for i in range(len(self.spec)):
            m = self.net._modules[str(2 * i)]
            out += [m.weight, m.bias]
        return out

    def packed(self):
        """Compute-dtype GEMM operands of every layer: w_fwd [Cout][tap][Cin_pad], w_bwd [Cin_pad][tap][Cout_pad], bias.
        Re-packed only when a parameter changed (optimizer step / load)."""
        ps = self.layer_params()
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if key == self._pack_key:
            return self._packed
        ce = chunk_elems(self.dtype)
        out = []
        with torch.no_grad():
            for i, (kind, cin, cout, k, s, pad, op) in enumerate(self.spec):
                w, b = ps[2 * i], ps[2 * i + 1]
                w_std = w if kind == "conv" else w.permute(1, 0, 2, 3)          # [Cout,Cin,kh,kw] view
                cin_pad, cout_pad = pad_to(cin, ce), pad_to(cout, ce)
                wf = torch.zeros((cout, k * k, cin_pad), dtype=torch.float32, device=w.device)
                wf[:, :, :cin] = w_std.permute(0, 2, 3, 1).reshape(cout, k * k, cin)
                wb = torch.zeros((cin_pad, k * k, cout_pad), dtype=torch.float32, device=w.device)
                wb[:cin, :, :cout] = w_std.permute(1, 2, 3, 0).reshape(cin, k * k, cout)
                out.append((wf.to(self.dtype).contiguous(), wb.to(self.dtype).contiguous(), b.detach().float().contiguous()))
        self._pack_key, self._packed = key, out
        return out

    def run(self, x_nhwc):
        return _StackFn.apply(x_nhwc, self, *self.layer_params())


class _ToNchw(torch.autograd.Function):
    """NHWC (padded pitch) -> fp32 NCHW; the backward re-pads with zeros so padded channels carry no gradient."""

    @staticmethod
    def forward(ctx, y, C):
        ctx.dt, ctx.pitch = y.dtype, y.shape[3]
        return ops.nhwc_to_nchw(y, C)

    @staticmethod
    def backward(ctx, g):
        return ops.nchw_to_nhwc(g.contiguous().float(), ctx.dt, ctx.pitch), None


class Encoder(_Stack):
    def __init__(self, input_dim=3, output_dim=256, image_size=100, dtype=torch.float32, device="cuda:0"):
        if image_size != 100 or input_dim != 3 or output_dim != 256:
            raise NotImplementedError("only the image_size=100 configuration used by the shipped run scripts is built")
        super().__init__(ENCODER_100, dtype, device)

    def forward(self, x):
        """x: fp32 [B,S,3,100,100] in [0,1] (or uint8 NHWC frames [B,S,100,100,3]) -> fp32 [B,S,256].
        Frames are data: no gradient is produced for x."""
        dev = self.device
        with torch.no_grad():
            if x.dtype == torch.uint8:
                B, S = x.shape[:2]
                frames = x.reshape(B * S, *x.shape[2:]).to(dev).contiguous()
                h = ops.u8_to_nhwc(frames, self.dtype, chunk_elems(self.dtype))      # [-1,1]; SLAC wants [0,1]:
                h = (h.float() + 1.0).mul_(0.5).to(self.dtype)
                h[..., 3:] = 0
            else:
                B, S, C, H, W = x.shape
                h = ops.nchw_to_nhwc(x.reshape(B * S, C, H, W).to(dev, torch.float32).contiguous(), self.dtype,
                                     chunk_elems(self.dtype))
        y = self.run(h)                                            # [B*S,1,1,256]
        return y.reshape(B, S, -1).float()


class Decoder(_Stack):
    def __init__(self, input_dim=288, output_dim=3, std=1.0, image_size=100, dtype=torch.float32, device="cuda:0"):
        if image_size != 100 or input_dim != 288 or output_dim != 3:
            raise NotImplementedError("only the image_size=100 configuration used by the shipped run scripts is built")
        super().__init__(DECODER_100, dtype, device)
        self.std = std

    def forward(self, z):
        """z: fp32 [B,S,288] -> (mean fp32 [B,S,3,100,100], std tensor filled with self.std).  Differentiable in z."""
        B, S, L = z.shape
        h = z.to(self.device).reshape(B * S, 1, 1, L).to(self.dtype).contiguous()
        y = self.run(h)                                            # [B*S,100,100,pitch]
        img = _ToNchw.apply(y, 3).reshape(B, S, 3, y.shape[1], y.shape[2])
        return img, torch.full_like(img, self.std)

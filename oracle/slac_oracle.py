"""CPU oracle for the SLAC encoder / decoder conv stacks -- TEST INFRASTRUCTURE ONLY.

Restates `/root/reference/rlkit/torch/slac/network/latent.py`: Decoder (:55-113, image_size 100 branch :82-101) and
Encoder (:116-171, image_size 100 branch :141-160).  PINNED by tests/golden/slac_golden_v1.npz (forward) and
tests/golden/slac_bwd_golden_v1.npz (parameter / latent gradients): outputs of the REAL reference modules loaded with
the seeded weights of `make_params` (tests/golden/make_golden_slac.py, make_golden_slac_bwd.py).
"""
import torch
import torch.nn.functional as F

# (kind, cin, cout, k, stride, pad, output_padding) -- every layer is followed by LeakyReLU(0.2)
ENCODER_100 = [("conv", 3, 32, 5, 2, 2, 0), ("conv", 32, 64, 3, 2, 1, 0), ("conv", 64, 128, 3, 2, 1, 0),
               ("conv", 128, 256, 3, 2, 1, 0), ("conv", 256, 256, 3, 2, 1, 0), ("conv", 256, 256, 4, 1, 0, 0)]
DECODER_100 = [("convT", 288, 256, 4, 1, 0, 0), ("convT", 256, 256, 3, 2, 1, 0), ("convT", 256, 128, 3, 2, 1, 0),
               ("convT", 128, 64, 3, 2, 1, 0), ("convT", 64, 32, 3, 2, 1, 1), ("convT", 32, 3, 5, 2, 2, 1)]


def make_params(spec, seed):
    """Seeded xavier-scaled weights + small random biases under the reference's state_dict keys (net.<2i>.weight/bias)."""
    g = torch.Generator().manual_seed(seed)
    p = {}
    for i, (kind, cin, cout, k, s, pad, op) in enumerate(spec):
        shape = (cout, cin, k, k) if kind == "conv" else (cin, cout, k, k)
        bound = (6.0 / ((cin + cout) * k * k)) ** 0.5
        p[f"net.{2 * i}.weight"] = (torch.rand(shape, generator=g) * 2 - 1) * bound
        p[f"net.{2 * i}.bias"] = torch.randn(cout, generator=g) * 0.05
    return p


def run_stack(spec, p, x):
    for i, (kind, cin, cout, k, s, pad, op) in enumerate(spec):
        w, b = p[f"net.{2 * i}.weight"], p[f"net.{2 * i}.bias"]
        if kind == "conv":
            x = F.conv2d(x, w, b, stride=s, padding=pad)
        else:
            x = F.conv_transpose2d(x, w, b, stride=s, padding=pad, output_padding=op)
        x = F.leaky_relu(x, 0.2)
    return x


def encoder_forward(p, x):
    """x [B,S,C,H,W] -> [B,S,256]   (latent.py:164-171)"""
    B, S, C, H, W = x.shape
    return run_stack(ENCODER_100, p, x.reshape(B * S, C, H, W)).reshape(B, S, -1)


def decoder_forward(p, z):
    """z [B,S,288] -> mean image [B,S,3,100,100]   (latent.py:105-113; the std output is a constant)"""
    B, S, L = z.shape
    y = run_stack(DECODER_100, p, z.reshape(B * S, L, 1, 1))
    return y.reshape(B, S, *y.shape[1:])


def backward_case(seed_in, seed_r):
    """Seeded inputs and upstream gradients of the backward fixture (shared by the fixture generator and the tests):
    frames x [2,2,3,100,100] in [0,1] on the uint8 grid, latents z [2,2,288], d(loss)/d(feat), d(loss)/d(img)."""
    g = torch.Generator().manual_seed(seed_in)
    x = (torch.rand(2, 2, 3, 100, 100, generator=g) * 255).round() / 255.0
    z = torch.randn(2, 2, 288, generator=g)
    g = torch.Generator().manual_seed(seed_r)
    r_feat = torch.randn(2, 2, 256, generator=g)
    r_img = torch.randn(2, 2, 3, 100, 100, generator=g) * 0.05
    return x, z, r_feat, r_img

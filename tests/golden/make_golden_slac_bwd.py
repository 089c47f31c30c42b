"""Generates tests/golden/slac_bwd_golden_v1.npz by RUNNING THE REAL REFERENCE Encoder / Decoder
(`/root/reference/rlkit/torch/slac/network/latent.py`, importable in the build container only) forward AND backward
with the seeded weights of oracle/slac_oracle.make_params.  Data only: inputs, upstream gradients (as seeds), and a
compact image of every parameter gradient (sum, L2 norm, a strided sample) plus the full latent gradient.
Run:  python tests/golden/make_golden_slac_bwd.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
sys.path.insert(0, "/root/reference")
import slac_oracle as SO  # noqa: E402
from rlkit.torch.slac.network.latent import Decoder, Encoder  # noqa: E402  (the real reference)

SEED_E, SEED_D, SEED_IN, SEED_R = 811, 812, 199, 77
NSAMP = 256


def sample(g):
    f = g.detach().double().flatten()
    stride = max(1, f.numel() // NSAMP)
    return f[::stride][:NSAMP].numpy()


def main():
    torch.set_num_threads(4)
    enc, dec = Encoder(3, 256, 100), Decoder(288, 3, 1.0, 100)
    pe, pd = SO.make_params(SO.ENCODER_100, SEED_E), SO.make_params(SO.DECODER_100, SEED_D)
    enc.load_state_dict(pe); dec.load_state_dict(pd)
    x, z, r_feat, r_img = SO.backward_case(SEED_IN, SEED_R)
    z = z.clone().requires_grad_(True)
    feat = enc(x)
    (feat * r_feat).sum().backward()
    img, _ = dec(z)
    (img * r_img).sum().backward()
    out = dict(seeds=np.array([SEED_E, SEED_D, SEED_IN, SEED_R]), dz=z.grad.numpy(),
               feat=feat.detach().numpy(), img_sum=img.detach().double().sum((3, 4)).numpy())
    for name, mod in (("enc", enc), ("dec", dec)):
        for k, v in mod.state_dict(keep_vars=True).items():
            g = v.grad
            out[f"{name}.{k}.sum"] = np.float64(g.double().sum().item())
            out[f"{name}.{k}.l2"] = np.float64(g.double().norm().item())
            out[f"{name}.{k}.samp"] = sample(g)
    path = os.path.join(HERE, "slac_bwd_golden_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

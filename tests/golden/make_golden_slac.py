"""Generates tests/golden/slac_golden_v1.npz by RUNNING THE REAL REFERENCE Encoder / Decoder
(`/root/reference/rlkit/torch/slac/network/latent.py`, importable in the build container only) with the seeded
weights of oracle/slac_oracle.make_params.  Data only: inputs, reference outputs, parameter checksums.
Run:  python tests/golden/make_golden_slac.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
sys.path.insert(0, "/root/reference")
import slac_oracle as SO  # noqa: E402
from rlkit.torch.slac.network.latent import Decoder, Encoder  # noqa: E402  (the real reference)

SEED_E, SEED_D = 811, 812


def main():
    torch.set_num_threads(4)
    enc, dec = Encoder(3, 256, 100), Decoder(288, 3, 1.0, 100)
    pe, pd = SO.make_params(SO.ENCODER_100, SEED_E), SO.make_params(SO.DECODER_100, SEED_D)
    enc.load_state_dict(pe); dec.load_state_dict(pd)
    g = torch.Generator().manual_seed(99)
    x = torch.rand(2, 2, 3, 100, 100, generator=g)            # frames in [0,1] as the SLAC buffer feeds them (buffer.py:135)
    z = torch.randn(2, 2, 288, generator=g)
    with torch.no_grad():
        feat = enc(x)
        img, std = dec(z)
    cs = lambda p: float(sum(v.double().abs().sum() for v in p.values()))
    out = dict(x=(x * 255).round().to(torch.uint8).numpy(), z=z.numpy(), feat=feat.numpy(), img=img[:, :, :, ::4, ::4].numpy(),
               img_sum=img.double().sum((3, 4)).numpy(), std_const=np.float32(std.flatten()[0].item()),
               checksum=np.array([cs(pe), cs(pd)]), seeds=np.array([SEED_E, SEED_D]))
    # the reference was run on the uint8-quantised frames so the fixture can store them compactly
    with torch.no_grad():
        out["feat"] = enc(torch.from_numpy(out["x"]).float() / 255.0).numpy()
    path = os.path.join(HERE, "slac_golden_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

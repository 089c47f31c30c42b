"""N3 -- SLAC encoder / decoder conv stacks.  PINNED parity: tests/golden/slac_golden_v1.npz holds outputs of the REAL
reference modules (rlkit/torch/slac/network/latent.py) run with the seeded weights of oracle/slac_oracle.make_params."""
import os

import numpy as np
import pytest
import torch

import slac_oracle as SO

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "slac_golden_v1.npz"))


def close(a, b, tol):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12)) < tol


def _params():
    pe, pd = SO.make_params(SO.ENCODER_100, int(G["seeds"][0])), SO.make_params(SO.DECODER_100, int(G["seeds"][1]))
    cs = [float(sum(v.double().abs().sum() for v in p.values())) for p in (pe, pd)]
    assert np.allclose(cs, G["checksum"], rtol=1e-9)            # identical weights to the ones the reference ran with
    return pe, pd


def test_oracle_restatement_matches_the_real_reference():
    pe, pd = _params()
    x = torch.from_numpy(G["x"]).float() / 255.0
    with torch.no_grad():
        feat = SO.encoder_forward(pe, x)
        img = SO.decoder_forward(pd, torch.from_numpy(G["z"]))
    assert feat.shape == (2, 2, 256) and close(feat, G["feat"], 1e-5)
    assert img.shape == (2, 2, 3, 100, 100)
    assert close(img[:, :, :, ::4, ::4], G["img"], 1e-5) and close(img.double().sum((3, 4)), G["img_sum"], 1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 4e-2)])
def test_hip_slac_stacks_match_the_real_reference(hip_device, dtype, tol):
    from s2p_amd.slac import Decoder, Encoder
    pe, pd = _params()
    enc = Encoder(3, 256, 100, dtype=dtype).load_state_dict(pe)
    dec = Decoder(288, 3, 1.0, 100, dtype=dtype).load_state_dict(pd)
    x = torch.from_numpy(G["x"]).float() / 255.0
    feat = enc(x).cpu()
    assert feat.shape == (2, 2, 256) and close(feat, G["feat"], tol)
    img, std = dec(torch.from_numpy(G["z"]))
    img = img.cpu()
    assert img.shape == (2, 2, 3, 100, 100) and float(std.flatten()[0]) == float(G["std_const"])
    assert close(img[:, :, :, ::4, ::4], G["img"], tol) and close(img.double().sum((3, 4)), G["img_sum"], tol)

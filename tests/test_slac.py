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
    feat = enc(x).detach().cpu()
    assert feat.shape == (2, 2, 256) and close(feat, G["feat"], tol)
    img, std = dec(torch.from_numpy(G["z"]))
    img = img.detach().cpu()
    assert img.shape == (2, 2, 3, 100, 100) and float(std.flatten()[0]) == float(G["std_const"])
    assert close(img[:, :, :, ::4, ::4], G["img"], tol) and close(img.double().sum((3, 4)), G["img_sum"], tol)


# ---- backward (training) ---------------------------------------------------------------------------------------
GB = np.load(os.path.join(HERE, "golden", "slac_bwd_golden_v1.npz"))


def _sample(g):
    f = torch.as_tensor(g).detach().double().flatten().cpu()
    stride = max(1, f.numel() // 256)
    return f[::stride][:256]


def _check_grads(name, named_grads, tol):
    for k, g in named_grads.items():
        ref_l2 = float(GB[f"{name}.{k}.l2"])
        got = torch.as_tensor(g).detach().double().cpu()
        assert abs(float(got.norm()) - ref_l2) <= tol * ref_l2, (name, k, float(got.norm()), ref_l2)
        assert abs(float(got.sum()) - float(GB[f"{name}.{k}.sum"])) <= tol * (ref_l2 * got.numel() ** 0.5), (name, k)
        samp = torch.from_numpy(GB[f"{name}.{k}.samp"])
        assert float((_sample(got) - samp).norm()) <= tol * max(float(samp.norm()), 1e-3 * ref_l2), (name, k)


def test_oracle_backward_matches_the_real_reference():
    pe, pd = (SO.make_params(SO.ENCODER_100, int(GB["seeds"][0])), SO.make_params(SO.DECODER_100, int(GB["seeds"][1])))
    x, z, r_feat, r_img = SO.backward_case(int(GB["seeds"][2]), int(GB["seeds"][3]))
    for p in list(pe.values()) + list(pd.values()):
        p.requires_grad_(True)
    z = z.clone().requires_grad_(True)
    feat = SO.encoder_forward(pe, x)
    (feat * r_feat).sum().backward()
    img = SO.decoder_forward(pd, z)
    (img * r_img).sum().backward()
    assert close(feat.detach(), GB["feat"], 1e-5) and close(z.grad, GB["dz"], 1e-5)
    _check_grads("enc", {k: v.grad for k, v in pe.items()}, 1e-5)
    _check_grads("dec", {k: v.grad for k, v in pd.items()}, 1e-5)


@pytest.mark.gpu
# fp32 is exact to 1e-5 in practice (tolerance 1e-3).  bf16 (bf16 operands and bf16 inter-layer gradients, fp32 accumulate)
# drifts by ~1.2 % relative L2 per layer walked backwards -- 0.3 % at the last layer, 7 % at the first / at dz
# (tests/tools/diag_slac_bwd.py) -- hence the 0.1 bound; the reference itself trains these stacks in fp32.
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 1e-1)])
def test_hip_slac_backward_matches_the_real_reference(hip_device, dtype, tol):
    from s2p_amd.slac import Decoder, Encoder
    pe, pd = (SO.make_params(SO.ENCODER_100, int(GB["seeds"][0])), SO.make_params(SO.DECODER_100, int(GB["seeds"][1])))
    x, z, r_feat, r_img = SO.backward_case(int(GB["seeds"][2]), int(GB["seeds"][3]))
    enc = Encoder(3, 256, 100, dtype=dtype).load_state_dict(pe)
    dec = Decoder(288, 3, 1.0, 100, dtype=dtype).load_state_dict(pd)
    assert set(enc.state_dict().keys()) == set(pe.keys()) and set(dec.state_dict().keys()) == set(pd.keys())
    zc = z.cuda().requires_grad_(True)
    feat = enc(x)
    (feat * r_feat.cuda()).sum().backward()
    img, _ = dec(zc)
    (img * r_img.cuda()).sum().backward()
    torch.cuda.synchronize()
    assert close(feat.detach().cpu(), GB["feat"], tol if dtype == torch.float32 else 4e-2)
    dz_ref = torch.from_numpy(GB["dz"]).double()
    dz_err = float((zc.grad.cpu().double() - dz_ref).norm() / dz_ref.norm())      # relative L2 (bf16: 6 layers of bf16 operands)
    assert dz_err < tol, dz_err
    _check_grads("enc", {k: v.grad for k, v in enc.state_dict(keep_vars=True).items()}, tol)
    _check_grads("dec", {k: v.grad for k, v in dec.state_dict(keep_vars=True).items()}, tol)
    # one Adam step through torch.optim moves the parameters and invalidates the packed operands
    opt = torch.optim.Adam(list(enc.parameters()) + list(dec.parameters()), lr=1e-3)
    before = enc(x).detach().clone()
    opt.step()
    assert float((enc(x).detach() - before).abs().max()) > 0

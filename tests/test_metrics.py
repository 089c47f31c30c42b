"""N4 -- PSNR / SSIM on device.  Parity UNPINNED (no reference code or numbers, `rebuttal.md:50` names them only):
the oracle restates the published definitions; these tests check the oracle against closed-form known answers and the
HIP kernel against the oracle."""
import numpy as np
import pytest
import torch

import metrics_oracle as MO


def test_oracle_known_answers():
    rng = np.random.default_rng(0)
    a = rng.uniform(-1, 1, (2, 3, 40, 33))
    assert np.allclose(MO.ssim(a, a), 1.0) and np.all(np.isinf(MO.psnr(a, a)))
    # constant offset d: MSE = d^2 -> PSNR = 20 log10(R/d); SSIM of a constant pair reduces to the luminance term
    d = 0.1
    assert np.allclose(MO.psnr(a, a + d), 20 * np.log10(2.0 / d))
    c0, c1 = np.full((1, 1, 20, 20), 0.25), np.full((1, 1, 20, 20), 0.5)
    lum = (2 * 0.25 * 0.5 + 0.02 ** 2) / (0.25 ** 2 + 0.5 ** 2 + 0.02 ** 2)
    assert np.allclose(MO.ssim(c0, c1), lum)
    g = MO.gaussian_window()
    assert len(g) == 11 and abs(g.sum() - 1) < 1e-15 and np.allclose(g, g[::-1]) and g.argmax() == 5
    # symmetry and range
    b = rng.uniform(-1, 1, a.shape)
    assert np.allclose(MO.ssim(a, b), MO.ssim(b, a)) and np.all(np.abs(MO.ssim(a, b)) <= 1)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(4, 3, 84, 84), (2, 3, 100, 100), (3, 1, 37, 53), (1, 3, 11, 11), (2, 3, 256, 256)])
def test_hip_metrics_match_oracle(hip_device, shape):
    from s2p_amd import metrics
    g = torch.Generator().manual_seed(sum(shape))
    a = torch.rand(shape, generator=g) * 2 - 1
    b = (a + 0.15 * torch.randn(shape, generator=g)).clamp(-1, 1)
    p, s = metrics.image_metrics(a.cuda(), b.cuda())
    assert p.shape == (shape[0],) and s.shape == (shape[0],)
    assert np.allclose(p.cpu().numpy(), MO.psnr(a.numpy(), b.numpy()), rtol=1e-4)
    assert np.allclose(s.cpu().numpy(), MO.ssim(a.numpy(), b.numpy()), rtol=1e-4, atol=1e-5)
    p2, s2 = metrics.image_metrics(a.cuda(), a.cuda())
    assert torch.isinf(p2).all() and np.allclose(s2.cpu().numpy(), 1.0, atol=1e-5)


@pytest.mark.gpu
def test_hip_metrics_reject_bad_input(hip_device):
    from s2p_amd import metrics
    with pytest.raises(RuntimeError):
        metrics.image_metrics(torch.zeros(1, 3, 10, 30).cuda(), torch.zeros(1, 3, 10, 30).cuda())      # smaller than the window
    with pytest.raises(ValueError):
        metrics.image_metrics(torch.zeros(1, 3, 20, 20).cuda(), torch.zeros(1, 3, 20, 21).cuda())
    with pytest.raises(RuntimeError):
        metrics.image_metrics(torch.zeros(1, 3, 20, 20), torch.zeros(1, 3, 20, 20))                    # no CPU fallback

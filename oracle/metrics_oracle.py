"""CPU oracle for the image-fidelity metrics -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: `/root/reference` names PSNR / SSIM as the paper's fidelity metrics (`rebuttal.md:50`) but holds no
code, call site, test or number for them.  This file restates the published definitions:
  PSNR = 10 log10(R^2 / MSE)
  SSIM (Wang, Bovik, Sheikh, Simoncelli 2004): 11x11 Gaussian window, sigma 1.5, weights normalised to 1,
       C1 = (0.01 R)^2, C2 = (0.03 R)^2, evaluated on fully covered window positions ("valid" filtering), mean over
       channels and positions.
float64 throughout."""
import numpy as np


def gaussian_window(size=11, sigma=1.5):
    d = np.arange(size, dtype=np.float64) - (size - 1) / 2
    g = np.exp(-d * d / (2 * sigma * sigma))
    return g / g.sum()


def _filt(x, g):
    """valid separable filtering of [N,C,H,W] along H and W"""
    k = len(g)
    H, W = x.shape[2], x.shape[3]
    t = sum(g[i] * x[:, :, :, i:W - k + 1 + i] for i in range(k))
    return sum(g[i] * t[:, :, i:H - k + 1 + i, :] for i in range(k))


def psnr(a, b, data_range=2.0):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    mse = ((a - b) ** 2).mean(axis=(1, 2, 3))
    with np.errstate(divide="ignore"):
        return 10.0 * np.log10(data_range * data_range / mse)


def ssim(a, b, data_range=2.0):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    g = gaussian_window()
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    mu_a, mu_b = _filt(a, g), _filt(b, g)
    va, vb, cov = _filt(a * a, g) - mu_a ** 2, _filt(b * b, g) - mu_b ** 2, _filt(a * b, g) - mu_a * mu_b
    m = ((2 * mu_a * mu_b + c1) * (2 * cov + c2)) / ((mu_a ** 2 + mu_b ** 2 + c1) * (va + vb + c2))
    return m.mean(axis=(1, 2, 3))

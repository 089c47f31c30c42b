"""Image-fidelity metrics on device: PSNR and SSIM between generated and ground-truth
frames, the fidelity metrics the paper reports (`rebuttal.md:50`; the reference holds no code or numbers for them, so
the definition is the published one -- see oracle/metrics_oracle.py, parity unpinned).  One fused HIP pass over fp32
NCHW batches, per-image results; no host round trip."""
import math

import torch

from . import _lib
from ._lib import check, lib, ptr, stream


def _prep(a, b):
    if a.shape != b.shape or a.dim() != 4:
        raise ValueError("metrics expect two NCHW batches of the same shape, got %s and %s" % (tuple(a.shape), tuple(b.shape)))
    if a.device.type != "cuda" or b.device.type != "cuda":
        raise RuntimeError("s2p_amd.metrics runs on a HIP device only (no CPU fallback)")
    return a.float().contiguous(), b.float().contiguous()


def image_metrics(a, b, data_range=2.0):
    """a, b: NCHW frames (default range [-1,1] -> data_range 2).  Returns (psnr [N], ssim [N]) fp32 device tensors.
    psnr is +inf for identical images.  The kernel uses no LDS: it may run on any stream, also beside
    training launches."""
    a, b = _prep(a, b)
    N, C, H, W = a.shape
    acc = torch.zeros((2, N), dtype=torch.float32, device=a.device)
    check(lib().s2p_image_metrics(ptr(a), ptr(b), N, C, H, W, float(data_range), ptr(acc[0]), ptr(acc[1]), stream()),
          "s2p_image_metrics")
    mse = acc[0] / float(C * H * W)
    psnr = 10.0 * torch.log10((data_range * data_range) / mse)
    ssim = acc[1] / float(C * (H - 10) * (W - 10))
    return psnr, ssim


def psnr(a, b, data_range=2.0):
    return image_metrics(a, b, data_range)[0]


def ssim(a, b, data_range=2.0):
    return image_metrics(a, b, data_range)[1]

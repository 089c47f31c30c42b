"""N-step autoregressive generation I_{t+1} = G(I_t, s_{t+1}) (README.md:30 `--seq_len`).  Frames stay resident in
HBM in the kernels' NHWC compute layout between steps: no host round trip and no layout conversion per step."""
import torch

from . import ops
from ._lib import chunk_elems


@torch.no_grad()
def rollout(netG, image0, states):
    """image0: fp32 NCHW [B,3,H,W]; states: fp32 [B,T,S] (s_{t+1} for t=0..T-1) -> fp32 NCHW frames [B,T,3,H,W]
    (a [T,B,...] buffer viewed batch-first: every step's frames are one contiguous block for the layout kernel)."""
    dt = netG.compute_dtype
    dev = netG.store.master.device
    img = ops.nchw_to_nhwc(image0.to(dev, torch.float32).contiguous(), dt, chunk_elems(dt))
    states = states.to(dev, torch.float32)
    B, T = states.shape[0], states.shape[1]
    # time-major storage: s2p_nhwc_to_nchw writes a CONTIGUOUS [B,3,H,W] block (out[:, t] of a batch-major buffer is strided
    # for B > 1 -- frames of different samples overwrote each other there)
    out = torch.empty((T, B, 3, img.shape[1], img.shape[2]), dtype=torch.float32, device=dev)
    for t in range(T):
        img, _ = netG.fwd_nhwc(img, states[:, t].contiguous(), save=False)
        ops.nhwc_to_nchw(img, 3, out=out[t])
    return out.transpose(0, 1)

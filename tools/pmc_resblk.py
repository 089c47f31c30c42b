"""Runs the dominant launch (ResBlk 3x3 256->256 conv, bs 64, 21x21, bf16: forward, dgrad, wgrad) a few times --
the target program for rocprofv3 --pmc passes (see profiles/round1_pmc_*.md)."""
import math, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
sys.path.insert(1, os.path.join(R, "tools")); import uselib  # noqa: E402  (S2P_LIB=<second build> for an A/B)
import torch
from s2p_amd import ops
dev = torch.device("cuda:0"); dt = torch.bfloat16
N, H, W, C = 64, 21, 21, 256
geom = ops.ConvGeom(C, C, 3, 1, 1)
x = torch.randn(N, H, W, C, device=dev).to(dt)
wf = (torch.randn(1, C, 9, C, device=dev) / math.sqrt(C * 9)).to(dt)
wb = (torch.randn(1, C, 9, C, device=dev) / math.sqrt(C * 9)).to(dt)
dy = torch.randn(N, H, W, C, device=dev).to(dt)
dw = torch.zeros(C * 9 * C, device=dev)
for _ in range(20):
    ops.conv_fwd(geom, x, wf, None, C)
    ops.conv_dgrad(geom, dy, wb, tuple(x.shape), C)
    ops.conv_wgrad(geom, x, dy, dw, C, C, C)
torch.cuda.synchronize()
print("done")

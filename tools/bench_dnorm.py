"""PatchGAN InstanceNorm + LeakyReLU launches (plain IN, small planes), forward and backward, us per call from a hipGraph
over rotating buffers (nothing stays cache-resident).  A/B: S2P_LIB=.../libs2p_hip_diag.so [S2P_NORM_NO_SMALL=1 | S2P_NORM_NO_512=1]."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
sys.path.insert(1, os.path.join(R, "tools")); import uselib  # noqa: E402  (S2P_LIB=<second build> for an A/B)
import torch
from s2p_amd import ops
from s2p_amd._lib import ACT_LRELU
dev = torch.device("cuda:0"); dt = torch.bfloat16
K = 8


def timeit(fn, n=K):
    for i in range(n): fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n): fn(i)
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (10 * n) * 1e3


tot_f = tot_b = 0.0
for (N, H, W, C, per_step) in [(64, 22, 22, 128, 5), (64, 21, 21, 256, 1), (64, 12, 12, 256, 5), (64, 13, 13, 512, 5), (64, 12, 12, 128, 5), (64, 7, 7, 256, 5), (64, 8, 8, 512, 5)]:
    xs = [torch.randn(N, H, W, C, device=dev).to(dt) for _ in range(K)]
    ds = [torch.randn(N, H, W, C, device=dev).to(dt) for _ in range(K)]
    _, st = ops.in_norm_fwd(xs[0], C, act=ACT_LRELU, slope=0.2)
    tf = timeit(lambda i: ops.in_norm_fwd(xs[i], C, act=ACT_LRELU, slope=0.2))
    tb = timeit(lambda i: ops.in_bwd(ds[i], xs[i], C, st, act=ACT_LRELU, slope=0.2))
    mb = N * H * W * C * 2 / 1e6
    print("N %d %2dx%-2d C %3d: fwd %5.1f us (%.2f TB/s) | bwd %5.1f us (%.2f TB/s)" % (N, H, W, C, tf, 2 * mb / tf, tb, 3 * mb / tb), flush=True)
    tot_f += tf * per_step; tot_b += tb * per_step
print("weighted per step: fwd %.0f us, bwd %.0f us" % (tot_f, tot_b))

"""PatchGAN logit head (512 -> 1, 4x4) backward at the production batch: dgrad + InstanceNorm backward through s2p_conv2d_dgrad_mat (two
launches for this shape) and as the two separate calls, and the weight gradient (streaming form vs the round-2 kernel); us per call
from a hipGraph.
    S2P_LIB=.../libs2p_hip_diag.so S2P_DIAG_SET="12=1" python tools/bench_dhead.py     (switch 12 selects the round-2 weight-gradient kernel)"""
import math, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
sys.path.insert(1, os.path.join(R, "tools")); import uselib  # noqa: E402
import torch
from s2p_amd import ops
from s2p_amd._lib import ACT_LRELU, EPI_ADD, EPI_STORE
dev = torch.device("cuda:0"); dt = torch.bfloat16


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3


for (N, H) in [(128, 13), (128, 8), (64, 13), (64, 8)]:
    C = 512
    geom = ops.ConvGeom(C, 1, 4, 1, 2)
    Ho, Wo = geom.out_hw(H, H)
    xn = torch.randn(N, H, H, C, device=dev).to(dt)
    tap = torch.randn(N, H, H, C, device=dev).to(dt)
    dy = torch.zeros(N, Ho, Wo, 8, device=dev, dtype=dt); dy[..., 0] = torch.randn(N, Ho, Wo, device=dev).to(dt)
    wb = torch.zeros(1, C, 16, 8, device=dev, dtype=dt); wb[..., 0] = (torch.randn(1, C, 16, device=dev) / math.sqrt(C * 16)).to(dt)
    _, st = ops.in_norm_fwd(xn, C, act=ACT_LRELU, slope=0.2)
    t_f = timeit(lambda: ops.conv_dgrad_mat(geom, dy, wb, xn, C, st, None, 0, None, 0, ACT_LRELU, 0.2, None, 0, None, 0, aux=tap))
    t_d = timeit(lambda: ops.conv_dgrad(geom, dy, wb, (N, H, H, C), C, aux=tap, epi=EPI_ADD))
    dm = ops.conv_dgrad(geom, dy, wb, (N, H, H, C), C, aux=tap, epi=EPI_ADD)
    t_n = timeit(lambda: ops.in_bwd(dm, xn, C, st, act=ACT_LRELU, slope=0.2))
    dw = torch.zeros(1, 16, C, device=dev); db = torch.zeros(1, device=dev)
    t_w = timeit(lambda: ops.conv_wgrad_batched(geom, [(xn, 0, dy, 0, dw, db)], C, C, 1))
    print("N %3d  %2dx%-2d | dgrad_mat (one call) %.1f us | dgrad %.1f + norm bwd %.1f = %.1f | wgrad (+ reduce) %.1f us" % (
        N, H, H, t_f, t_d, t_n, t_d + t_n, t_w), flush=True)

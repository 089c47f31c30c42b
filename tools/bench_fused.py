"""ResBlk conv + MAT norm: separate launches vs the fused entry point (s2p_conv2d_fwd_mat), us per call from a hipGraph."""
import math, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
sys.path.insert(1, os.path.join(R, "tools")); import uselib  # noqa: E402  (S2P_LIB=<second build> for an A/B)
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


N, C = 64, 256
geom = ops.ConvGeom(C, C, 3, 1, 1)
x = torch.randn(N, 21, 21, C, device=dev).to(dt)
wf = (torch.randn(1, C, 9, C, device=dev) / math.sqrt(C * 9)).to(dt)
b = torch.randn(C, device=dev)
gb = (torch.randn(N, 21, 21, 12 * 2 * C, device=dev) * 0.3).to(dt)
st = torch.randn(N, 12 * 2 * C, device=dev) * 0.3
y = ops.conv_fwd(geom, x, wf, b, C)
t_conv = timeit(lambda: ops.conv_fwd(geom, x, wf, b, C))
t_conv_res = timeit(lambda: ops.conv_fwd(geom, x, wf, b, C, aux=x, epi=EPI_ADD))
t_norm = timeit(lambda: ops.in_norm_fwd(y, C, gb, 512, st, 512, ACT_LRELU, 0.2))
t_fused = timeit(lambda: ops.conv_fwd_mat(geom, x, wf, b, C, gb, 512, st, 512, ACT_LRELU, 0.2))
t_fused_res = timeit(lambda: ops.conv_fwd_mat(geom, x, wf, b, C, gb, 512, st, 512, ACT_LRELU, 0.2, aux=x, epi=EPI_ADD))
print("conv %.1f us | conv+skip %.1f | norm %.1f | conv -> norm (two launches) %.1f | fused %.1f | fused + skip %.1f" % (
    t_conv, t_conv_res, t_norm, t_conv + t_norm, t_fused, t_fused_res))
# backward: dgrad + norm backward, separate vs fused
wb = (torch.randn(1, C, 9, C, device=dev) / math.sqrt(C * 9)).to(dt)
dyv = torch.randn(N, 21, 21, C, device=dev).to(dt)
ym, stats = ops.in_norm_fwd(x, C, gb, 512, st, 512, ACT_LRELU, 0.2)
dgb = torch.empty_like(gb); dst = torch.empty_like(st)
t_dg = timeit(lambda: ops.conv_dgrad(geom, dyv, wb, tuple(x.shape), C))
dm = ops.conv_dgrad(geom, dyv, wb, tuple(x.shape), C)
t_nb = timeit(lambda: ops.in_bwd(dm, x, C, stats, gb, 512, st, 512, ACT_LRELU, 0.2, dgb, 512, dst, 512))
t_fb = timeit(lambda: ops.conv_dgrad_mat(geom, dyv, wb, x, C, stats, gb, 512, st, 512, ACT_LRELU, 0.2, dgb, 512, dst, 512))
t_fbr = timeit(lambda: ops.conv_dgrad_mat(geom, dyv, wb, x, C, stats, gb, 512, st, 512, ACT_LRELU, 0.2, dgb, 512, dst, 512, res=x))
print("dgrad %.1f us | norm bwd %.1f | two launches %.1f | fused %.1f | fused + skip gradient %.1f" % (t_dg, t_nb, t_dg + t_nb, t_fb, t_fbr))

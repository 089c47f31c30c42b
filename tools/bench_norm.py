"""Micro-benchmark of the InstanceNorm / MAT kernels at the train-step shapes (bf16). Usage: python tools/bench_norm.py"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(1, os.path.join(R, "tools")); import uselib  # noqa: E402  (S2P_LIB=<second build> for an A/B)
import torch
from s2p_amd import ops
from s2p_amd._lib import ACT_LRELU, ACT_RELU
dev = torch.device("cuda:0"); dt = torch.bfloat16
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (N, H, W, C, mod) in [(64, 21, 21, 256, True), (64, 84, 84, 64, False), (64, 42, 42, 128, False), (128, 22, 22, 128, False), (128, 12, 12, 256, False)]:
    x = torch.randn(N, H, W, C, device=dev).to(dt)
    gb = torch.randn(N, H, W, 12 * 2 * C if mod else 8, device=dev).to(dt) if mod else None
    st = torch.randn(N, 12 * 2 * C, device=dev) if mod else None
    da = torch.randn(N, H, W, C, device=dev).to(dt)
    dgb = torch.empty_like(gb) if mod else None
    act = ACT_LRELU if mod else ACT_RELU
    el = N * H * W * C * 2 / 1e6
    t_s = timeit(lambda: ops.in_stats(x, C))
    stats = ops.in_stats(x, C)
    t_f = timeit(lambda: ops.in_apply_fwd(x, C, stats, gb, 2 * C, st, 2 * C, act, 0.2))
    t_b = timeit(lambda: ops.in_bwd(da, x, C, stats, gb, 2 * C, st, 2 * C, act, 0.2, dgb, 2 * C))
    fb = el * (4 if mod else 2); bb = el * ((4 + 4 + 3) if mod else (3 + 4))
    print("[%d,%d,%d,%d] mod=%d  stats %5.1f us (%.2f TB/s)  apply fwd %5.1f us (%.2f TB/s)  bwd(reduce+apply) %5.1f us (%.2f TB/s)"
          % (N, H, W, C, mod, t_s, el / t_s, t_f, fb / t_f, t_b, bb / t_b))
# fused single-launch forms at the ResBlk shape, graph-timed, next to an elementwise add of the same tensors (streaming rate)
def gtime(fn, n=20):
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
N, H, W, C = 64, 21, 21, 256
x = torch.randn(N, H, W, C, device=dev).to(dt); da = torch.randn(N, H, W, C, device=dev).to(dt)
gb = torch.randn(N, H, W, 12 * 2 * C, device=dev).to(dt); st = torch.randn(N, 12 * 2 * C, device=dev)
dgb = torch.empty_like(gb); dst = torch.zeros(N, 12 * 2 * C, device=dev)
el = N * H * W * C * 2 / 1e6
t = gtime(lambda: ops.in_norm_fwd(x, C, gb, 2 * C, st, 2 * C, ACT_LRELU, 0.2))
print("fused MAT fwd   %5.1f us  (%.2f TB/s of %d MB)" % (t, 4 * el / t, 4 * el))
t = gtime(lambda: ops.in_norm_fwd(x, C, None, 0, None, 0, ACT_RELU, 0.2))
print("fused IN  fwd   %5.1f us  (%.2f TB/s of %d MB)" % (t, 2 * el / t, 2 * el))
y, stats = ops.in_norm_fwd(x, C, gb, 2 * C, st, 2 * C, ACT_LRELU, 0.2)
t = gtime(lambda: ops.in_bwd(da, x, C, stats, gb, 2 * C, st, 2 * C, ACT_LRELU, 0.2, dgb, 2 * C, dst, 2 * C))
print("fused MAT bwd   %5.1f us  (%.2f TB/s of %d MB)" % (t, 7 * el / t, 7 * el))
t = gtime(lambda: ops.in_stats(x, C)); print("stats only      %5.1f us  (%.2f TB/s)" % (t, el / t))
t = gtime(lambda: ops.in_apply_fwd(x, C, stats, gb, 2 * C, st, 2 * C, ACT_LRELU, 0.2)); print("apply only      %5.1f us  (%.2f TB/s)" % (t, 4 * el / t))
o = torch.empty_like(x)
t = gtime(lambda: ops.add(x, da, out=o)); print("add (2r+1w)     %5.1f us  (%.2f TB/s)" % (t, 3 * el / t))

"""ResBlk chain as ONE chain over the whole batch vs TWO half-batch chains on two streams: does a
chain's HBM-bound norm tail / launch boundary hide under the other chain's MFMA loop?  12 fused conv -> MAT-norm launches
(forward) and 12 fused dgrad -> norm-backward launches, each reading the previous one's output, with twelve DISTINCT gamma|beta
planes (cold, as in the real step); us per chain from a hipGraph."""
import math, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
sys.path.insert(1, os.path.join(R, "tools")); import uselib  # noqa: E402  (S2P_LIB=<second build> for an A/B)
import torch
from s2p_amd import ops
from s2p_amd._lib import ACT_LRELU, EPI_ADD, EPI_STORE
dev = torch.device("cuda:0"); dt = torch.bfloat16
N, C, K = 64, 256, 12
geom = ops.ConvGeom(C, C, 3, 1, 1)
x0 = torch.randn(N, 21, 21, C, device=dev).to(dt)
wf = [(torch.randn(1, C, 9, C, device=dev) / math.sqrt(C * 9)).to(dt) for _ in range(K)]
b = torch.randn(C, device=dev)
gb = (torch.randn(K, N, 21, 21, 2 * C, device=dev) * 0.3).to(dt)      # group-major planes, 347 MB
st = torch.randn(N, K * 2 * C, device=dev) * 0.3
dgb = torch.empty_like(gb); dst = torch.empty_like(st)
s2 = torch.cuda.Stream()


def chain_fwd(lo, hi):
    x = x0[lo:hi]
    for k in range(K):
        y, x, _ = ops.conv_fwd_mat(geom, x, wf[k], b, C, gb[k][lo:hi], 0, st[lo:hi], k * 2 * C, ACT_LRELU, 0.2)
    return x


_, stats_full = ops.in_norm_fwd(x0, C, gb[0], 0, st, 0, ACT_LRELU, 0.2)
_, stats_h = ops.in_norm_fwd(x0[:N // 2], C, gb[0][:N // 2], 0, st[:N // 2], 0, ACT_LRELU, 0.2)


def chain_bwd(lo, hi, stats):
    d = x0[lo:hi]
    for k in range(K):
        d = ops.conv_dgrad_mat(geom, d, wf[k], x0[lo:hi], C, stats, gb[k][lo:hi], 0, st[lo:hi], k * 2 * C, ACT_LRELU, 0.2,
                               dgb[k][lo:hi], 0, dst[lo:hi], k * 2 * C)
    return d


def dual(fn):
    def run():
        main = torch.cuda.current_stream()
        s2.wait_stream(main)
        with torch.cuda.stream(s2):
            fn(N // 2, N)
        fn(0, N // 2)
        main.wait_stream(s2)
    return run


def timeit(fn, n=4):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="relaxed"):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(ts)[len(ts) // 2]


for rnd in range(3):
    a = timeit(lambda: chain_fwd(0, N)); bb = timeit(dual(chain_fwd))
    c = timeit(lambda: chain_bwd(0, N, stats_full)); d = timeit(dual(lambda lo, hi: chain_bwd(lo, hi, stats_h)))
    print("round %d: forward chain of %d fused launches: one chain %.1f us (%.1f per launch) | two half-batch chains %.1f us (%.1f) || "
          "backward: one chain %.1f us (%.1f) | two %.1f us (%.1f)" % (rnd, K, a, a / K, bb, bb / K, c, c / K, d, d / K), flush=True)

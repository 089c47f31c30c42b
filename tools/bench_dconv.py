"""PatchGAN conv layers (forward / dgrad), us per call from a hipGraph over rotating buffers.
A/B of the K-split plan: S2P_LIB=.../libs2p_hip_diag.so S2P_SPLIT_MIN_STEPS=8 ..."""
import math, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
sys.path.insert(1, os.path.join(R, "tools")); import uselib  # noqa: E402  (S2P_LIB=<second build> for an A/B)
import torch
from s2p_amd import ops
from s2p_amd._lib import ACT_LRELU
dev = torch.device("cuda:0"); dt = torch.bfloat16
if len(sys.argv) > 1:          # k=v ...: diagnostics switches
    import ctypes
    from s2p_amd import _lib
    for kv in sys.argv[1:]:
        k_, v_ = kv.split("="); assert ctypes.CDLL(_lib._SO).s2p_diag_set(int(k_), int(v_)) == 0
K = 6


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


tf_ = tb_ = 0.0
for (N, H, W, ci, co, k, s, p) in [(64, 43, 43, 64, 128, 4, 2, 2), (64, 22, 22, 128, 256, 4, 2, 2), (64, 12, 12, 256, 512, 4, 1, 2),
                                   (64, 22, 22, 64, 128, 4, 2, 2), (64, 12, 12, 128, 256, 4, 2, 2), (64, 7, 7, 256, 512, 4, 1, 2),
                                   (64, 10, 10, 512, 512, 3, 1, 1), (64, 5, 5, 512, 512, 3, 1, 1), (64, 84, 84, 64, 64, 3, 1, 1),
                                   (64, 42, 42, 64, 128, 3, 1, 1), (64, 42, 42, 128, 128, 3, 1, 1)]:
    geom = ops.ConvGeom(ci, co, k, s, p)
    Ho, Wo = geom.out_hw(H, W)
    xs = [torch.randn(N, H, W, ci, device=dev).to(dt) for _ in range(K)]
    dys = [torch.randn(N, Ho, Wo, co, device=dev).to(dt) for _ in range(K)]
    wf = (torch.randn(1, co, k * k, ci, device=dev) / math.sqrt(ci * k * k)).to(dt)
    wb = (torch.randn(1, ci, k * k, co, device=dev) / math.sqrt(ci * k * k)).to(dt)
    b = torch.randn(co, device=dev)
    tf = timeit(lambda i: ops.conv_fwd(geom, xs[i], wf, b, ci))
    tb = timeit(lambda i: ops.conv_dgrad(geom, dys[i], wb, (N, H, W, ci), ci))
    gf = 2.0 * N * Ho * Wo * co * ci * k * k / 1e9
    print("(%d,%d,%d,%d->%d,k%d,s%d): fwd %5.1f us %4.0f TF | dgrad %5.1f us %4.0f TF" % (N, H, W, ci, co, k, s, tf, gf / tf * 1e3, tb, gf / tb * 1e3), flush=True)
    tf_ += tf; tb_ += tb
print("sum: fwd %.0f us, dgrad %.0f us" % (tf_, tb_))

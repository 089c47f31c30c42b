"""Plain InstanceNorm + ReLU forward on the encoder / decoder planes (84x84 x 64, 42x42 x 128 channels, N 64): register-resident
single launch against reduce + apply (diagnostics switch 9), us per call over rotating buffers, and the largest difference between the
two results.   S2P_LIB=s2p_amd/csrc/libs2p_hip_diag.so python tools/bench_norm_big.py"""
import os, sys, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
sys.path.insert(1, os.path.join(R, "tools")); import uselib  # noqa: E402  (S2P_LIB=<second build> for an A/B)
import torch
from s2p_amd import ops, _lib
from s2p_amd._lib import ACT_RELU
raw = ctypes.CDLL(_lib._SO)
dev = torch.device("cuda:0"); K = 6


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


for (N, H, W, C) in [(64, 84, 84, 64), (64, 42, 42, 128), (64, 43, 43, 64), (3, 30, 37, 192)]:
    xs = [(torch.randn(N, H, W, C, device=dev) * 1.7 + 0.4).bfloat16() for _ in range(K)]
    out = {}
    for sw in (0, 1):
        assert raw.s2p_diag_set(9, sw) == 0
        y, s = ops.in_norm_fwd(xs[0], C, act=ACT_RELU)
        torch.cuda.synchronize()
        out[sw] = (y.float().clone(), timeit(lambda i: ops.in_norm_fwd(xs[i], C, act=ACT_RELU)))
    mb = 2 * N * H * W * C * 2 / 1e6
    print("[%d,%d,%d,%d]: fused %6.1f us (%.2f TB/s) | reduce + apply %6.1f us | max |diff| %.3g" %
          (N, H, W, C, out[0][1], mb / out[0][1], out[1][1], float((out[0][0] - out[1][0]).abs().max())), flush=True)

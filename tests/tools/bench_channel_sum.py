"""Bias-gradient pass (s2p_channel_sum with its scratch: fixed-order partial sums) at the train-step sizes, us per call."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
sys.path.insert(1, os.path.join(R, "tools")); import uselib  # noqa: E402  (S2P_LIB=<second build> for an A/B)
import torch
from s2p_amd import ops
dev = torch.device("cuda:0"); K = 4
for (N, H, W, C) in [(64, 21, 21, 1536), (64, 84, 84, 3), (64, 13, 13, 512), (128, 43, 43, 64)]:
    xs = [torch.randn(N, H, W, ops.pad_to(C, 8), device=dev).bfloat16() for _ in range(K)]
    db = torch.zeros(C, device=dev)
    for i in range(K): ops.channel_sum(xs[i], C, db)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(K): ops.channel_sum(xs[i], C, db)
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / (10 * K) * 1e3
    ref = xs[0].float().sum((0, 1, 2))[:C]
    db.zero_(); ops.channel_sum(xs[0], C, db); torch.cuda.synchronize()
    print("[%d,%d,%d,%d]: %6.1f us  %.2f TB/s  rel err %.2e" % (N, H, W, C, t, xs[0].numel() * 2 / t / 1e6, float((db - ref).abs().max() / ref.abs().max())), flush=True)

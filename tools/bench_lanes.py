"""Does splitting the batch into two half-batches on two streams help the ResBlk chain?  One graph with 24 dependent
3x3 256->256 convs (+ optional norms) at N=64 on one stream  vs  the same chain at N=32 on each of two streams: per-CU the
two lanes are out of phase, so one lane's prologue / epilogue can overlap the other's MFMA loop.
Usage: python tools/bench_lanes.py [with_norm]"""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2p_amd import ops

dev = torch.device("cuda:0"); dt = torch.bfloat16
H = W = 21; C = 256; L = 24
with_norm = len(sys.argv) > 1
geom = ops.ConvGeom(C, C, 3, 1, 1)
wf = (torch.randn(1, C, 9, C, device=dev) / math.sqrt(C * 9)).to(dt)


def chain(x, ys):
    cur = x
    for i in range(L):
        if with_norm:
            cur, _ = ops.in_norm_fwd(cur, C, None, 0, None, 0, 1, 0.2)
        ops.conv_fwd(geom, cur, wf, None, C, out=ys[i & 1])
        cur = ys[i & 1]
    return cur


def capture(nl):
    N = 64 // nl
    xs = [torch.randn(N, H, W, C, device=dev).to(dt) for _ in range(nl)]
    ys = [[torch.empty(N, H, W, C, device=dev, dtype=dt) for _ in range(2)] for _ in range(nl)]
    streams = [torch.cuda.Stream() for _ in range(nl)]
    for l in range(nl):
        chain(xs[l], ys[l])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        for l in range(nl):
            streams[l].wait_stream(main)
            with torch.cuda.stream(streams[l]):
                chain(xs[l], ys[l])
        for l in range(nl):
            main.wait_stream(streams[l])
    return g


for nl in (1, 2, 4):
    try:
        g = capture(nl)
    except TypeError as e:
        print("conv_fwd has no out= parameter:", e); break
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    fl = 2.0 * 64 * H * W * C * C * 9 * L
    print("lanes=%d  %8.1f us per chain of %d  (%.1f us per layer, %.0f TFLOP/s conv-only)" % (nl, us, L, us / L, fl / us / 1e6), flush=True)

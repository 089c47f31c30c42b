"""Thin-input forward convs at the train-step sizes (VGG conv1_1, conditioning conv 3 -> 1536, stem, PatchGAN first layers), us per call."""
import math, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
sys.path.insert(1, os.path.join(R, "tools")); import uselib  # noqa: E402  (S2P_LIB=<second build> for an A/B)
import torch
from s2p_amd import ops
from s2p_amd._lib import ACT_RELU
dev = torch.device("cuda:0"); dt = torch.bfloat16; K = 4
for (N, H, W, ci, co, k, s, p, refl) in [(64, 84, 84, 3, 64, 3, 1, 1, False), (64, 21, 21, 3, 1536, 3, 1, 1, False), (64, 84, 84, 3, 64, 7, 1, 3, True),
                                         (64, 84, 84, 6, 64, 4, 2, 2, False), (64, 42, 42, 6, 64, 4, 2, 2, False)]:
    geom = ops.ConvGeom(ci, co, k, s, p, reflect=refl)
    Ho, Wo = geom.out_hw(H, W)
    xs = [torch.randn(N, H, W, 8, device=dev).to(dt) for _ in range(K)]
    for x in xs: x[..., ci:] = 0
    wf = (torch.randn(co, k * k, 8, device=dev) / math.sqrt(ci * k * k)).to(dt)
    b = torch.randn(co, device=dev)
    fn = lambda i: ops.conv_fwd(geom, xs[i], wf, b, 8, act=ACT_RELU)
    for i in range(K): fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(K): fn(i)
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / (10 * K) * 1e3
    print("(%d,%d,%d,%d->%d,k%d,s%d): %6.1f us  output %.2f TB/s" % (N, H, W, ci, co, k, s, t, N * Ho * Wo * co * 2 / t / 1e6), flush=True)

"""Weight gradients of the thin layers at the train-step sizes (stem 3 -> 64 7x7 reflect, output conv 64 -> 3 7x7 reflect, conditioning
conv 3 -> 1536 3x3, PatchGAN first layers 6 -> 64 4x4 stride 2), us per call incl. their reduce / bias passes."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
sys.path.insert(1, os.path.join(R, "tools")); import uselib  # noqa: E402  (S2P_LIB=<second build> for an A/B)
import torch
from s2p_amd import ops
dev = torch.device("cuda:0"); dt = torch.bfloat16; K = 3
for (N, H, W, ci, co, k, s, p, refl, bias) in [(64, 84, 84, 3, 64, 7, 1, 3, True, False), (64, 84, 84, 64, 3, 7, 1, 3, True, True),
                                               (64, 21, 21, 3, 1536, 3, 1, 1, False, True), (64, 84, 84, 6, 64, 4, 2, 2, False, True),
                                               (64, 42, 42, 6, 64, 4, 2, 2, False, True)]:
    geom = ops.ConvGeom(ci, co, k, s, p, reflect=refl)
    Ho, Wo = geom.out_hw(H, W)
    cip, cop = ops.pad_to(ci, 8), ops.pad_to(co, 8)
    xs = [torch.randn(N, H, W, cip, device=dev).to(dt) for _ in range(K)]
    dys = [torch.randn(N, Ho, Wo, cop, device=dev).to(dt) for _ in range(K)]
    dw = torch.zeros(co, k * k, ci, device=dev); db = torch.zeros(co, device=dev) if bias else None
    fn = lambda i: ops.conv_wgrad(geom, xs[i], dys[i], dw, cip, ci, co, db=db)
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
    gf = 2.0 * N * Ho * Wo * co * ci * k * k / 1e9
    print("(%d,%d,%d,%d->%d,k%d,s%d): %6.1f us %5.0f TF" % (N, H, W, ci, co, k, s, t, gf / t * 1e3), flush=True)

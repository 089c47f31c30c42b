"""Weight gradients of the strided / 4x4 layers at the train-step shapes, us per call (hipGraph over rotating buffers).
A/B against the implicit-GEMM kernel:  S2P_LIB=s2p_amd/csrc/libs2p_hip_diag.so python tools/bench_wgrad_strided.py [ab]
(`ab`: every shape also with diagnostics switch 5 = padded-raster kernel off)."""
import math, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
sys.path.insert(1, os.path.join(R, "tools")); import uselib  # noqa: E402  (S2P_LIB=<second build> for an A/B)
import torch
from s2p_amd import ops
from s2p_amd import _lib
import ctypes
_raw = ctypes.CDLL(_lib._SO)
dev = torch.device("cuda:0"); dt = torch.bfloat16
K = 4


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


ab = "ab" in sys.argv[1:]
for kv in sys.argv[1:]:          # k=v: diagnostics switch k := v before anything is timed
    if "=" in kv:
        k_, v_ = kv.split("="); assert _raw.s2p_diag_set(int(k_), int(v_)) == 0
tot = [0.0, 0.0]
for (N, H, W, ci, co, k, s, p, tr) in [(64, 84, 84, 64, 128, 3, 2, 1, False), (64, 42, 42, 128, 256, 3, 2, 1, False),
                                       (64, 21, 21, 256, 128, 3, 2, 1, True), (64, 42, 42, 128, 64, 3, 2, 1, True),
                                       (128, 43, 43, 64, 128, 4, 2, 2, False), (128, 22, 22, 128, 256, 4, 2, 2, False),
                                       (128, 12, 12, 256, 512, 4, 1, 2, False), (128, 22, 22, 64, 128, 4, 2, 2, False),
                                       (128, 12, 12, 128, 256, 4, 2, 2, False), (128, 7, 7, 256, 512, 4, 1, 2, False)]:
    geom = ops.ConvGeom(ci, co, k, s, p, transposed=tr, output_padding=1 if tr else 0)
    Ho, Wo = geom.out_hw(H, W)
    xs = [torch.randn(N, H, W, ci, device=dev).to(dt) for _ in range(K)]
    dys = [torch.randn(N, Ho, Wo, co, device=dev).to(dt) for _ in range(K)]
    rows, cols = (ci, co) if tr else (co, ci)
    dw = torch.zeros(rows, k * k, cols, device=dev)
    gf = 2.0 * N * (H * W if tr else Ho * Wo) * co * ci * k * k / 1e9
    res = []
    for sw in ([0, 1] if ab else [0]):
        if ab: assert _raw.s2p_diag_set(5, sw) == 0
        t = timeit(lambda i: ops.conv_wgrad(geom, xs[i], dys[i], dw, ci, ci, co))
        res.append(t); tot[sw] += t
    print("(%d,%d,%d,%d->%d,k%d,s%d%s): " % (N, H, W, ci, co, k, s, ",T" if tr else "") +
          " | ".join("%6.1f us %4.0f TF" % (t, gf / t * 1e3) for t in res), flush=True)
print("sum: " + " | ".join("%.0f us" % t for t in (tot if ab else tot[:1])))

"""State-path linear layers (csrc/linear_small.hip), us per call from a hipGraph over rotating buffers."""
import math, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
sys.path.insert(1, os.path.join(R, "tools")); import uselib  # noqa: E402  (S2P_LIB=<second build> for an A/B)
import torch
from s2p_amd import ops
from s2p_amd._lib import ACT_LRELU, ACT_NONE
dev = torch.device("cuda:0")
Kb = 6


def timeit(fn, n=Kb):
    for i in range(n): fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n): fn(i)
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (20 * n) * 1e3


for (M, K, N, lrelu) in [(64, 360, 256, True), (64, 256, 256, True), (64, 256, 6144, False)]:
    xs = [torch.randn(M, K, device=dev) for _ in range(Kb)]
    wf = torch.randn(1, N, 1, K, device=dev) / math.sqrt(K); wb = wf[0, :, 0].t().contiguous().view(1, K, 1, N)
    b = torch.randn(N, device=dev); dy = torch.randn(M, N, device=dev)
    act = ACT_LRELU if lrelu else ACT_NONE
    y = ops.linear_fwd(xs[0], wf, b, K, N, act, 0.2)
    dw = torch.zeros(N, K, device=dev); db = torch.zeros(N, device=dev)
    tf = timeit(lambda i: ops.linear_fwd(xs[i], wf, b, K, N, act, 0.2))
    tb = timeit(lambda i: ops.linear_bwd(xs[i], dy, y if lrelu else None, wb, K, K, N, act, 0.2, dw, db))
    print("M %d K %d N %d: fwd %.1f us | bwd (wgrad + dgrad) %.1f us" % (M, K, N, tf, tb), flush=True)

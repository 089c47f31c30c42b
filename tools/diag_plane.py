"""Timing ablations of the plane-resident conv kernel (diagnostics build: S2P_LIB=.../libs2p_hip_diag.so, S2P_DIAG=n).
Prints us per launch of the ResBlk conv (64 x 21 x 21, 256 -> 256) and of Cin = 64 / 128 variants (fixed vs per-step cost)."""
import math, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from s2p_amd import ops
dev = torch.device("cuda:0"); dt = torch.bfloat16


def timeit(fn, n=20):
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


out = []
for N, cin, cout in ((64, 256, 256), (64, 128, 256), (64, 64, 256), (64, 256, 64), (128, 256, 256)):
    geom = ops.ConvGeom(cin, cout, 3, 1, 1)
    x = torch.randn(N, 21, 21, cin, device=dev).to(dt)
    wf = (torch.randn(1, cout, 9, cin, device=dev) / math.sqrt(cin * 9)).to(dt)
    t = timeit(lambda: ops.conv_fwd(geom, x, wf, None, cin, y_pitch=cout))
    fl = 2.0 * N * 441 * cin * cout * 9
    out.append("N=%d cin=%d cout=%d: %.1f us %.0f TF" % (N, cin, cout, t, fl / t / 1e6))
print("DIAG=%s  " % os.environ.get("S2P_DIAG", "0") + " | ".join(out), flush=True)
D = int(os.environ.get("S2P_DIAG", "0"))
if D & 48:      # in-kernel clock and per-segment cycles of the K loop (waves 0 and 4 of every workgroup)
    N, cin, cout = 64, 256, 256
    geom = ops.ConvGeom(cin, cout, 3, 1, 1)
    x = torch.randn(N, 21, 21, cin, device=dev).to(dt)
    wf = (torch.randn(1, cout, 9, cin, device=dev) / math.sqrt(cin * 9)).to(dt)
    for _ in range(200):                    # a few ms of back-to-back launches before the sample
        y = ops.conv_fwd(geom, x, wf, None, cin, y_pitch=cout)
    torch.cuda.synchronize()
    st = y.view(torch.int64).flatten()[:16 * N * cout // 64].view(-1, 2, 8).cpu().double()
    for s_ in (0, 1):
        cyc, rt = st[:, s_, 0], st[:, s_, 1]
        seg = st[:, s_, 2:8].median(0).values / 36
        print("loop set %d: median %.0f cycles, %.2f us, clock %.2f GHz; per pair-step %.0f cycles = reads %.0f + dma %.0f + mfma %.0f + topwait %.0f + barrier %.0f" % (
            s_, cyc.median(), rt.median() / 100, (cyc / rt * 0.1).median(), cyc.median() / 36, seg[0], seg[1], seg[2], seg[3], seg[4]))

"""Timing ablations of the plane-resident conv kernel (diagnostics build: S2P_LIB=.../libs2p_hip_diag.so, S2P_DIAG=n).
Prints us per launch of the ResBlk conv (64 x 21 x 21, 256 -> 256) and of Cin = 64 / 128 variants (fixed vs per-step cost)."""
import math, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
sys.path.insert(1, os.path.join(R, "tools")); import uselib  # noqa: E402  (S2P_LIB=<second build> for an A/B)
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
if D & 16:      # in-kernel clock of the K loop (wave 0 of every workgroup)
    N, cin, cout = 64, 256, 256
    geom = ops.ConvGeom(cin, cout, 3, 1, 1)
    x = torch.randn(N, 21, 21, cin, device=dev).to(dt)
    wf = (torch.randn(1, cout, 9, cin, device=dev) / math.sqrt(cin * 9)).to(dt)
    for _ in range(200):                    # a few ms of back-to-back launches before the sample
        y = ops.conv_fwd(geom, x, wf, None, cin, y_pitch=cout)
    torch.cuda.synchronize()
    st = y.view(torch.int64).flatten()[:2 * N * cout // 64].view(-1, 2).cpu().double()
    cyc, rt = st[:, 0], st[:, 1]
    print("loop: median %.0f cycles, %.2f us, clock %.2f GHz; %.0f cycles per pair-step (MFMA bound 896)" % (
        cyc.median(), rt.median() / 100, (cyc / rt * 0.1).median(), cyc.median() / 36))

if D & 128:     # phase time line of the complete kernel (realtime stamps -> aux)
    N, cin, cout = 64, 256, 256
    geom = ops.ConvGeom(cin, cout, 3, 1, 1)
    x = torch.randn(N, 21, 21, cin, device=dev).to(dt)
    wf = (torch.randn(1, cout, 9, cin, device=dev) / math.sqrt(cin * 9)).to(dt)
    dbg = torch.zeros(N * cout // 64 * 8, dtype=torch.int64, device=dev)
    for _ in range(200):
        y = ops.conv_fwd(geom, x, wf, None, cin, y_pitch=cout, aux=dbg)
    torch.cuda.synchronize()
    st = dbg.view(-1, 8).cpu().double()[:, :6] / 100.0          # us
    t0 = st[:, 0].min()
    names = ["entry", "loop start", "loop end", "merged", "staged", "stored"]
    print("phases (us after the first workgroup's entry; median / max over workgroups): " +
          ", ".join("%s %.2f/%.2f" % (names[k], (st[:, k] - t0).median(), (st[:, k] - t0).max()) for k in range(6)))

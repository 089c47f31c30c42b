"""In-kernel clock of the halo conv main loop (run with S2P_DIAG=8: the launch stamps s_memtime / s_memrealtime around
its K loop into the output buffer instead of storing results).  Prints the median shader clock, cycles per K step and
the MFMA-issue share of the loop.  Usage: S2P_DIAG=8 python tools/clock_halo.py"""
import math, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from s2p_amd import ops
assert os.environ.get("S2P_DIAG") == "8", "run with S2P_DIAG=8"
assert "diag" in os.environ.get("S2P_LIB", ""), "needs the diagnostics build: bash s2p_amd/csrc/build.sh diag; S2P_LIB=s2p_amd/csrc/libs2p_hip_diag.so"
dev = torch.device("cuda:0"); dt = torch.bfloat16
for (N, cin, label) in [(64, 256, "ResBlk conv bs64 (442 tiles)"), (74, 256, "510 tiles (2 per CU)"), (37, 256, "256 tiles (1 per CU)"), (74, 1024, "510 tiles, K=9216")]:
    cout = 256
    geom = ops.ConvGeom(cin, cout, 3, 1, 1)
    x = torch.randn(N, 21, 21, cin, device=dev).to(dt)
    wf = (torch.randn(1, cout, 9, cin, device=dev) / math.sqrt(cin * 9)).to(dt)
    y = torch.empty(N, 21, 21, cout, device=dev, dtype=dt)
    t0 = time.time()
    while time.time() - t0 < 2.5:                       # >= 2 s of back-to-back launches: DVFS steady state
        for _ in range(200): ops.conv_fwd(geom, x, wf, None, cin, y_pitch=cout, out=y)
        torch.cuda.synchronize()
    nblk = -(-N * 441 // 128) * 2
    st = y.view(torch.int64).flatten()[: nblk * 2].reshape(nblk, 2).cpu().double()
    cyc, real = st[:, 0], st[:, 1]
    clk = (cyc / real * 100e6).median().item()
    ksteps = cin // 64 * 9
    cps = (cyc / ksteps).median().item()
    print("%-30s clock %.2f GHz   %.0f cycles per K step per workgroup (16 MFMA 32x32x16 per wave = 512 MFMA cycles)   loop %.1f us"
          % (label, clk / 1e9, cps, (real.median().item() / 100e6) * 1e6), flush=True)

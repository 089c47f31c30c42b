"""Per-workgroup timeline of the pipelined halo conv (ResBlk 3x3 256->256 at [64,21,21]): s_memrealtime stamps at kernel
entry / K-loop start / K-loop end / after the epilogue (diagnostics build, S2P_DIAG=9; the launch computes normally).
    bash s2p_amd/csrc/build.sh diag && S2P_LIB=$PWD/s2p_amd/csrc/libs2p_hip_diag.so S2P_DIAG=9 python tools/timeline_halo.py"""
import math, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from s2p_amd import ops
assert os.environ.get("S2P_DIAG") == "9" and "diag" in os.environ.get("S2P_LIB", "")
dev = torch.device("cuda:0"); dt = torch.bfloat16
N, C = 64, 256
geom = ops.ConvGeom(C, C, 3, 1, 1)
x = torch.randn(N, 21, 21, C, device=dev).to(dt)
wf = (torch.randn(1, C, 9, C, device=dev) / math.sqrt(C * 9)).to(dt)
y = torch.empty(N, 21, 21, C, device=dev, dtype=dt)
nblk = -(-N * 441 // 128) * 2
dbg = torch.zeros(nblk * 4, dtype=torch.int64, device=dev)
for _ in range(300):
    ops.conv_fwd(geom, x, wf, None, C, y_pitch=C, out=y, aux=dbg)
torch.cuda.synchronize()
t = dbg.reshape(nblk, 4).cpu().double() * 0.01           # us
t0 = t[:, 0].min()
t = t - t0
q = lambda v: "min %.1f  med %.1f  p90 %.1f  max %.1f" % (v.min(), v.median(), v.quantile(0.9), v.max())
print("workgroups %d" % nblk)
print("entry (since first entry)   ", q(t[:, 0]))
print("prologue (entry -> loop)    ", q(t[:, 1] - t[:, 0]))
print("K loop                      ", q(t[:, 2] - t[:, 1]))
print("epilogue                    ", q(t[:, 3] - t[:, 2]))
print("workgroup total             ", q(t[:, 3] - t[:, 0]))
print("kernel span (first entry -> last exit) %.1f us" % t[:, 3].max())

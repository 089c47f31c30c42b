"""Diagnostic: per-kernel fp32 error against float64 torch references (run on the GPU box)."""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle"))
import torch, torch.nn.functional as F
from s2p_amd import ops
from s2p_amd._lib import ACT_LRELU, ACT_RELU, ACT_NONE, ACT_TANH
from test_kernels_gpu import nhwc, nchw, pack_fwd, pack_bwd

dev = torch.device("cuda:0")
def rel(a, b): return float((a.double() - b.double()).norm() / b.double().norm())
g = torch.Generator().manual_seed(0)
dt = torch.float32
# conv 3x3 256->256 on 21x21
N, C, H, W = 2, 256, 21, 21
x = torch.randn(N, C, H, W, generator=g); w = torch.randn(C, C, 3, 3, generator=g) / math.sqrt(C * 9); b = torch.randn(C, generator=g)
yr = F.conv2d(x.double(), w.double(), b.double(), padding=1)
geom = ops.ConvGeom(C, C, 3, 1, 1)
xd = nhwc(x, C, dt, dev)
y = ops.conv_fwd(geom, xd, pack_fwd(w, C, dt, dev), b.to(dev), C)
print("conv fwd fp32 rel-L2:", rel(nchw(y, C), yr), " torch-fp32:", rel(F.conv2d(x, w, b, padding=1), yr))
dy = torch.randn(N, C, H, W, generator=g)
dx = ops.conv_dgrad(geom, nhwc(dy, C, dt, dev), pack_bwd(w, C, C, dt, dev), tuple(xd.shape), C)
dxr = F.conv_transpose2d(dy.double(), w.double(), padding=1)
print("conv dgrad fp32:", rel(nchw(dx, C), dxr))
dw = torch.zeros(C, 9, C, device=dev)
ops.conv_wgrad(geom, xd, nhwc(dy, C, dt, dev), dw, C, C, C)
xr = x.double().requires_grad_(True); wr = w.double().requires_grad_(True)
F.conv2d(xr, wr, padding=1).backward(dy.double())
print("conv wgrad fp32:", rel(dw.cpu().reshape(C, 3, 3, C).permute(0, 3, 1, 2), wr.grad))
# instance norm modulated
x = torch.randn(N, C, H, W, generator=g) * 2 + 0.5
gam = torch.randn(N, C, H, W, generator=g) * 0.5; bet = torch.randn(N, C, H, W, generator=g) * 0.5
st = torch.randn(N, 2 * C, generator=g) * 0.5; da = torch.randn(N, C, H, W, generator=g)
xr, gr, br, sr = [t.double().requires_grad_(True) for t in (x, gam, bet, st)]
yr = F.leaky_relu(F.instance_norm(xr, eps=1e-5) * (1 + gr + sr[:, :C, None, None]) + br + sr[:, C:, None, None], 0.2)
yr.backward(da.double())
xd = nhwc(x, C, dt, dev)
gb = torch.cat([nhwc(gam, C, dt, dev), nhwc(bet, C, dt, dev)], 3).contiguous()
stats = ops.in_stats(xd, C)
y = ops.in_apply_fwd(xd, C, stats, gb, 0, st.to(dev), 0, ACT_LRELU, 0.2)
print("IN fwd fp32:", rel(nchw(y, C), yr.detach()))
dgb = torch.empty_like(gb)
dst = torch.empty(N, 2 * C, device=dev)
dx = ops.in_bwd(nhwc(da, C, dt, dev), xd, C, stats, gb, 0, st.to(dev), 0, ACT_LRELU, 0.2, dgb, 0, dst, 0)
print("IN bwd dx:", rel(nchw(dx, C), xr.grad), " dgamma:", rel(nchw(dgb, C), gr.grad), " dbeta:", rel(nchw(dgb[..., C:], C), br.grad),
      " dst:", rel(dst.cpu(), sr.grad))
# tanh / act_bwd
o = torch.randn(4, 5, 5, 8, generator=g)
print("posenc:", end=" ")
import s2p_oracle as O
s = torch.randn(8, 17, generator=g)
pe = ops.posenc(s.to(dev), 10, 360).cpu()
print(rel(pe[:, :357], O.positional_encoding(s.double(), 10)), " torch-fp32:", rel(O.positional_encoding(s, 10), O.positional_encoding(s.double(), 10)))
# linear as 1x1 conv
xl = torch.randn(64, 357, generator=g); wl = torch.randn(256, 357, generator=g) / 19
geom = ops.ConvGeom(357, 256, 1)
xp = torch.zeros(64, 1, 1, 360); xp[:, 0, 0, :357] = xl
wp = torch.zeros(256, 1, 360); wp[:, 0, :357] = wl
yl = ops.conv_fwd(geom, xp.to(dev), wp.to(dev), None, 360)
print("linear fwd:", rel(yl.view(64, 256).cpu(), xl.double() @ wl.double().t()))

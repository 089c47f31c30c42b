"""Does the fused conv+norm tail run faster when its gamma|beta plane was written just before (MALL-hot)?
A: one 12-group gamma/beta conv up front (product), then the chain of 11 fused convs reading slices of the 347 MB buffer.
B: a one-group gamma/beta conv right before each fused conv, into a 29 MB buffer of its own.   us per chain, from a hipGraph."""
import math, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
sys.path.insert(1, os.path.join(R, "tools")); import uselib  # noqa: E402  (S2P_LIB=<second build> for an A/B)
import torch
from s2p_amd import ops
from s2p_amd._lib import ACT_LRELU, ACT_NONE
dev = torch.device("cuda:0"); dt = torch.bfloat16


def timeit(fn, n=4):
    for _ in range(2): fn()
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


N, C, NH, NN = 64, 256, 128, 12
geom = ops.ConvGeom(C, C, 3, 1, 1)
g12 = ops.ConvGeom(NH, 2 * C, 3, 1, 1, groups=NN, x_gstride=NH, y_gstride=2 * C)
g1 = ops.ConvGeom(NH, 2 * C, 3, 1, 1)
x0 = torch.randn(N, 21, 21, C, device=dev).to(dt)
wf = [(torch.randn(1, C, 9, C, device=dev) / math.sqrt(C * 9)).to(dt) for _ in range(NN)]
b = torch.randn(C, device=dev)
actv = torch.randn(N, 21, 21, NN * NH, device=dev).to(dt)
w12 = (torch.randn(NN, 2 * C, 9, NH, device=dev) / math.sqrt(NH * 9)).to(dt)
b12 = torch.randn(NN * 2 * C, device=dev) * 0.1
st = torch.randn(N, NN * 2 * C, device=dev) * 0.3
actv_k = [actv[..., k * NH:(k + 1) * NH].contiguous() for k in range(NN)]
w1 = [w12[k:k + 1].contiguous() for k in range(NN)]
b1 = [b12[k * 2 * C:(k + 1) * 2 * C].contiguous() for k in range(NN)]
st1 = [st[:, k * 2 * C:(k + 1) * 2 * C].contiguous() for k in range(NN)]


def chain_a():
    gb = ops.conv_fwd(g12, actv, w12, b12, NH)
    x = x0
    for k in range(1, NN):
        _, x, _ = ops.conv_fwd_mat(geom, x, wf[k], b, C, gb, k * 2 * C, st, k * 2 * C, ACT_LRELU, 0.2)
    return x


def chain_b():
    x = x0
    for k in range(1, NN):
        gb = ops.conv_fwd(g1, actv_k[k], w1[k], b1[k], NH)
        _, x, _ = ops.conv_fwd_mat(geom, x, wf[k], b, C, gb, 0, st1[k], 0, ACT_LRELU, 0.2)
    return x


g12n = ops.ConvGeom(NH, 2 * C, 3, 1, 1, groups=NN, x_gstride=NH, y_gstride=N * 21 * 21 * 2 * C)
gbn = torch.empty((NN, N, 21, 21, 2 * C), dtype=dt, device=dev)


def chain_n():
    """A with the planes norm-major ([12][N][HW][2C], rows of 1 KB) instead of pixel-major ([N][HW][12*2C], rows of 12 KB)"""
    ops.conv_fwd(g12n, actv, w12, b12, NH, y_pitch=2 * C, out=gbn)
    x = x0
    for k in range(1, NN):
        _, x, _ = ops.conv_fwd_mat(geom, x, wf[k], b, C, gbn[k], 0, st, k * 2 * C, ACT_LRELU, 0.2)
    return x


def chain_n_rev():
    """N consumed in the REVERSE of the production order (the planes written last are read first)"""
    ops.conv_fwd(g12n, actv, w12, b12, NH, y_pitch=2 * C, out=gbn)
    x = x0
    for k in range(NN - 1, 0, -1):
        _, x, _ = ops.conv_fwd_mat(geom, x, wf[k], b, C, gbn[k], 0, st, k * 2 * C, ACT_LRELU, 0.2)
    return x


def only_gb12n(): ops.conv_fwd(g12n, actv, w12, b12, NH, y_pitch=2 * C, out=gbn)


def only_gb12(): return ops.conv_fwd(g12, actv, w12, b12, NH)
def only_gb1():
    for k in range(1, NN): ops.conv_fwd(g1, actv_k[k], w1[k], b1[k], NH)


ya, yb = chain_a(), chain_b()
print("chains agree:", float((ya.float() - yb.float()).abs().max()))
ta, tb = timeit(chain_a), timeit(chain_b)
t12, t1 = timeit(only_gb12), timeit(only_gb1)
print("A (12-group up front) %.1f us | B (per norm) %.1f us | gamma/beta convs alone: 12-group %.1f, 11 x one-group %.1f" % (ta, tb, t12, t1))
yn = chain_n()
print("norm-major chain agrees:", float((ya.float() - yn.float()).abs().max()))
tn, t12n = timeit(chain_n), timeit(only_gb12n)
print("N (norm-major planes) %.1f us | its 12-group conv %.1f us | fused convs %.1f us each" % (tn, t12n, (tn - t12n) / 11))
tnr = timeit(chain_n_rev)
print("N consumed newest-first: %.1f us | fused convs %.1f us each" % (tnr, (tnr - t12n) / 11))
print("fused convs in A: %.1f us each | in B: %.1f us each" % ((ta - t12) / 11, (tb - t1) / 11))

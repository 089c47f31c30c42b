"""Sweep of the K-split count of the batched slab weight-gradient kernel (csrc/wgrad_slab.hip) on the two shapes of the
train step: 12 ResBlk convs 256->256 and the 12 gamma/beta groups 128->512, bs 64 at 21x21.  Needs the diagnostics build
(S2P_WGRAD_SLAB_SPLITS is only read there):
    bash s2p_amd/csrc/build.sh diag && S2P_LIB=$PWD/s2p_amd/csrc/libs2p_hip_diag.so python tools/bench_wgrad_slab.py"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
sys.path.insert(1, os.path.join(R, "tools")); import uselib  # noqa: E402  (S2P_LIB=<second build> for an A/B)
import torch
from s2p_amd import ops

dev = torch.device("cuda:0")
N, H, W = 64, 21, 21


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def case(name, nj, cin, cout, grouped):
    geom = ops.ConvGeom(cin, cout, 3, 1, 1)
    if grouped:
        x = torch.randn(N, H, W, nj * cin, device=dev).bfloat16(); dy = torch.randn(N, H, W, nj * cout, device=dev).bfloat16()
    else:
        xs = [torch.randn(N, H, W, cin, device=dev).bfloat16() for _ in range(nj)]
        dys = [torch.randn(N, H, W, cout, device=dev).bfloat16() for _ in range(nj)]
    dw = torch.zeros(nj, cout, 9, cin, device=dev); db = torch.zeros(nj, cout, device=dev)
    jobs = [(x, j * cin, dy, j * cout, dw[j], db[j]) if grouped else (xs[j], 0, dys[j], 0, dw[j], db[j]) for j in range(nj)]
    flops = 2.0 * N * H * W * cin * cout * 9 * nj
    for S in SLIST:
        os.environ["S2P_WGRAD_SLAB_SPLITS"] = str(S)
        us = timeit(lambda: ops.conv_wgrad_batched(geom, jobs, cin, cin, cout))
        print("%-10s S=%d  %8.1f us  %6.0f TFLOP/s" % (name, S, us, flops / us / 1e6), flush=True)


SLIST = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else (0, 1, 2, 3, 4, 6, 8)
which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "resblk"):
    case("resblk", 12, 256, 256, False)
if which in ("all", "gammabeta"):
    case("gammabeta", 12, 128, 512, True)

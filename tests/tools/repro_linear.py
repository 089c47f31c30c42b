"""Isolation test for the split-K dgrad of the state path's 256 -> 6144 affine layer (s2p_linear_bwd): repeat the same call on a
side stream, with and without a heavy kernel running on the main stream, and compare the results bit for bit."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
sys.path.insert(1, os.path.join(R, "tools")); import uselib  # noqa: E402  (S2P_LIB=<second build> for an A/B)
import torch
from s2p_amd import ops
M, K, N = 64, 256, 6144
g = torch.Generator().manual_seed(0)
x = torch.randn(M, K, generator=g).cuda(); dy = torch.randn(M, N, generator=g).cuda()
w_bwd = torch.randn(1, K, 1, N, generator=g).cuda().contiguous()
big = torch.randn(8192, 8192, device="cuda")
side = torch.cuda.Stream()
for load in (False, True):
    outs, dws = [], []
    for it in range(40):
        dw = torch.zeros(N * K, device="cuda"); db = torch.zeros(N, device="cuda")
        side.wait_stream(torch.cuda.current_stream())      # dw / db are ready
        if load:
            y = big @ big                                  # runs on the main stream WHILE the side stream works
            z = torch.relu(big)                            # + an HBM-bound kernel
        with torch.cuda.stream(side):
            dx = ops.linear_bwd(x, dy, None, w_bwd, K, K, N, 0, 0.0, dw, db)
        torch.cuda.current_stream().wait_stream(side)
        outs.append(dx.clone()); dws.append(dw.clone())
    torch.cuda.synchronize()
    bad = sum(not torch.equal(o, outs[0]) for o in outs); badw = sum(not torch.equal(o, dws[0]) for o in dws)
    ref = dy.double() @ w_bwd.view(K, N).double().t()
    err = max(float((o.double() - ref).norm() / ref.norm()) for o in outs)
    print("concurrent load %-5s: %d of 40 dx differ from the first, %d of 40 dw differ; worst dx error vs float64 %.2e" % (load, bad, badw, err))

"""Calibration only (not on the product path): vendor GEMM (hipBLASLt via torch.matmul, bf16) on the plain-GEMM
equivalents of the dominant conv shapes, to see what a tuned library reaches at these sizes on this chip."""
import torch
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K, label) in [(28224, 256, 2304, "ResBlk conv fwd as GEMM [pix x K] @ [K x Cout]"),
                         (256, 2304, 28224, "ResBlk conv wgrad as GEMM [Cout x pix] @ [pix x K]"),
                         (28224, 512, 1152, "one gamma/beta group fwd"),
                         (56448, 256, 2304, "VGG3 conv fwd (2N)"),
                         (8192, 8192, 8192, "large square (library sweet spot)")]:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16); b = torch.randn(K, N, device="cuda", dtype=torch.bfloat16)
    us = t(lambda: a @ b)
    print("%-52s M=%6d N=%5d K=%6d  %7.1f us  %6.0f TFLOP/s" % (label, M, N, K, us, 2.0 * M * N * K / us / 1e6), flush=True)

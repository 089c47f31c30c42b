"""l1_loss kernel timing at the VGG / feature-matching tap sizes (bf16)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
sys.path.insert(1, os.path.join(R, "tools")); import uselib  # noqa: E402  (S2P_LIB=<second build> for an A/B)
import torch
from s2p_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
loss = torch.zeros(1, device=dev)
for shape in [(64, 84, 84, 64), (64, 42, 42, 128), (64, 21, 21, 256), (64, 10, 10, 512), (64, 43, 43, 64), (64, 5, 5, 512)]:
    a = torch.randn(*shape, device=dev).bfloat16(); b = torch.randn(*shape, device=dev).bfloat16(); g = torch.empty_like(a)
    t = timeit(lambda: ops.l1_loss(a, b, 1.0 / a.numel(), loss, g))
    print("%-22s %7.1f us  %5.2f TB/s" % (shape, t, a.numel() * 6 / t / 1e6), flush=True)
# the five VGG taps in ONE launch (s2p_l1_loss_multi), as the perceptual loss calls it
shapes = [(64, 84, 84, 64), (64, 42, 42, 128), (64, 21, 21, 256), (64, 10, 10, 512), (64, 5, 5, 512)]
A = [torch.randn(*s, device=dev).bfloat16() for s in shapes]; B = [torch.randn(*s, device=dev).bfloat16() for s in shapes]
G = [torch.empty_like(a) for a in A]
lm = torch.zeros(5, device=dev)
jobs = [(A[i], B[i], 1.0 / A[i].numel(), lm[i:i + 1], G[i]) for i in range(5)]
t = timeit(lambda: ops.l1_loss_multi(jobs))
tot = sum(a.numel() for a in A) * 6
print("five VGG taps, one launch: %7.1f us  %5.2f TB/s" % (t, tot / t / 1e6))

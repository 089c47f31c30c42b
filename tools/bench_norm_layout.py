"""Does the channel pitch of the gamma/beta maps matter to the MAT norm kernels?  gb_all [N,h,w,12*2C] (pitch 6144: each
norm reads 2 x 512 B per pixel out of a 12-KB row) vs one contiguous [N,h,w,2C] tensor per norm.  Both variants walk over
12 different norms' maps so that nothing stays cache-resident between calls."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from s2p_amd import ops
from s2p_amd._lib import ACT_LRELU
dev = torch.device("cuda:0"); dt = torch.bfloat16
N, H, W, C, K = 64, 21, 21, 256, 12
xs = [torch.randn(N, H, W, C, device=dev).to(dt) for _ in range(K)]
das = [torch.randn(N, H, W, C, device=dev).to(dt) for _ in range(K)]
gb_wide = torch.randn(N, H, W, K * 2 * C, device=dev).to(dt)
gb_sep = torch.randn(K, N, H, W, 2 * C, device=dev).to(dt)
dgb_wide = torch.empty_like(gb_wide); dgb_sep = torch.empty_like(gb_sep)
st = torch.randn(N, K * 2 * C, device=dev); dst = torch.empty_like(st)


def run(wide):
    for k in range(K):
        if wide:
            y, s = ops.in_norm_fwd(xs[k], C, gb_wide, k * 2 * C, st, k * 2 * C, ACT_LRELU, 0.2)
            ops.in_bwd(das[k], xs[k], C, s, gb_wide, k * 2 * C, st, k * 2 * C, ACT_LRELU, 0.2, dgb_wide, k * 2 * C, dst, k * 2 * C)
        else:
            y, s = ops.in_norm_fwd(xs[k], C, gb_sep[k], 0, st, k * 2 * C, ACT_LRELU, 0.2)
            ops.in_bwd(das[k], xs[k], C, s, gb_sep[k], 0, st, k * 2 * C, ACT_LRELU, 0.2, dgb_sep[k], 0, dst, k * 2 * C)


for wide in (True, False, True, False):
    run(wide); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run(wide)
    e1.record(); torch.cuda.synchronize()
    print("%s: %.1f us per (fwd + bwd) norm" % ("pitch 6144" if wide else "contiguous per norm", e0.elapsed_time(e1) / 5 / K * 1e3))

"""Fixed-vs-per-K-step cost of the halo conv kernel: 3x3 Cout=256 @21x21, Cin sweep, 1 or 2 workgroups per CU."""
import sys, os, math
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from s2p_amd import ops
dev = torch.device("cuda:0"); dt = torch.bfloat16
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for N in (37, 74):
    for cin in (64, 256):
        cout = 256
        geom = ops.ConvGeom(cin, cout, 3, 1, 1)
        x = torch.randn(N, 21, 21, cin, device=dev).to(dt)
        wf = (torch.randn(1, cout, 9, cin, device=dev) / math.sqrt(cin * 9)).to(dt)
        t = timeit(lambda: ops.conv_fwd(geom, x, wf, None, cin, y_pitch=cout))
        blocks = -(-N * 441 // 128) * 2
        fl = 2.0 * N * 441 * cin * cout * 9
        print("N=%3d blocks=%4d cin=%4d ksteps=%3d  %6.1f us  %5.0f TF" % (N, blocks, cin, cin // 64 * 9, t, fl / t / 1e6), flush=True)

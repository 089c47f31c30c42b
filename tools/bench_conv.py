"""Per-shape conv microbenchmark (fwd / dgrad / wgrad) for the layer shapes of the S2P train step at bs=64.
Usage: python tools/bench_conv.py [filter]"""
import sys, os, math
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(1, os.path.join(R, "tools")); import uselib  # noqa: E402  (S2P_LIB=<second build> for an A/B)
import torch
from s2p_amd import ops
dev = torch.device("cuda:0"); dt = torch.bfloat16
# name, N, H, W, cin, cout, k, stride, pad, transposed, reflect, groups
SHAPES = [
    ("resblk 3x3 256->256 @21", 64, 21, 21, 256, 256, 3, 1, 1, 0, 0, 1),
    ("gb grouped 12x(128->512) @21", 64, 21, 21, 128, 512, 3, 1, 1, 0, 0, 12),
    ("down0 64->128 s2 @84", 64, 84, 84, 64, 128, 3, 2, 1, 0, 0, 1),
    ("down1 128->256 s2 @42", 64, 42, 42, 128, 256, 3, 2, 1, 0, 0, 1),
    ("up0 convT 256->128 @21", 64, 21, 21, 256, 128, 3, 2, 1, 1, 0, 1),
    ("up1 convT 128->64 @42", 64, 42, 42, 128, 64, 3, 2, 1, 1, 0, 1),
    ("stem 7x7 3->64 @84", 64, 84, 84, 3, 64, 7, 1, 3, 0, 1, 1),
    ("out 7x7 64->3 @84", 64, 84, 84, 64, 3, 7, 1, 3, 0, 1, 1),
    ("shared 3->1536 @21", 64, 21, 21, 3, 1536, 3, 1, 1, 0, 0, 1),
    ("D0 6->64 s2 @84 (2N)", 128, 84, 84, 6, 64, 4, 2, 2, 0, 0, 1),
    ("D1 64->128 s2 @43", 128, 43, 43, 64, 128, 4, 2, 2, 0, 0, 1),
    ("D2 128->256 s2 @22", 128, 22, 22, 128, 256, 4, 2, 2, 0, 0, 1),
    ("D3 256->512 s1 @12", 128, 12, 12, 256, 512, 4, 1, 2, 0, 0, 1),
    ("D4 512->1 s1 @13", 128, 13, 13, 512, 1, 4, 1, 2, 0, 0, 1),
    ("N64 D3 256->512 s1 @12", 64, 12, 12, 256, 512, 4, 1, 2, 0, 0, 1),
    ("N64 D3 256->512 s1 @7", 64, 7, 7, 256, 512, 4, 1, 2, 0, 0, 1),
    ("N64 D2 128->256 s2 @12", 64, 12, 12, 128, 256, 4, 2, 2, 0, 0, 1),
    ("N64 vgg4_2 512->512 @10", 64, 10, 10, 512, 512, 3, 1, 1, 0, 0, 1),
    ("N64 vgg5_1 512->512 @5", 64, 5, 5, 512, 512, 3, 1, 1, 0, 0, 1),
    ("vgg1_1 3->64 @84", 64, 84, 84, 3, 64, 3, 1, 1, 0, 0, 1),
    ("vgg1_2 64->64 @84 (2N)", 128, 84, 84, 64, 64, 3, 1, 1, 0, 0, 1),
    ("vgg2_2 128->128 @42", 128, 42, 42, 128, 128, 3, 1, 1, 0, 0, 1),
    ("vgg3_2 256->256 @21", 128, 21, 21, 256, 256, 3, 1, 1, 0, 0, 1),
    ("vgg4_2 512->512 @10", 128, 10, 10, 512, 512, 3, 1, 1, 0, 0, 1),
    ("vgg5_1 512->512 @5", 128, 5, 5, 512, 512, 3, 1, 1, 0, 0, 1),
]
flt = sys.argv[1] if len(sys.argv) > 1 else ""
def timeit(fn, n=20):
    """us per call, replayed from a hipGraph of n back-to-back calls (no host launch cost in the number)."""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * n) * 1e3   # us
print("%-32s %10s %10s %10s   (us | TFLOP/s)" % ("shape", "fwd", "dgrad", "wgrad"))
tot = [0, 0, 0]
for (name, N, H, W, cin, cout, k, s, p, tr, refl, G) in SHAPES:
    if flt and flt not in name: continue
    geom = ops.ConvGeom(cin, cout, k, s, p, transposed=bool(tr), reflect=bool(refl), groups=G, output_padding=1 if tr else 0,
                        x_gstride=cin if G > 1 else 0, y_gstride=cout if G > 1 else 0)
    cp, op = ops.pad_to(cin, 8), ops.pad_to(cout, 8)
    x = torch.randn(N, H, W, cp * G, device=dev).to(dt)
    Ho, Wo = geom.out_hw(H, W)
    wf = (torch.randn(G, cout, k * k, cp, device=dev) / math.sqrt(cin * k * k)).to(dt)
    wb = (torch.randn(G, cp, k * k, op, device=dev) / math.sqrt(cin * k * k)).to(dt)
    dy = torch.randn(N, Ho, Wo, op * G, device=dev).to(dt)
    dw = torch.zeros(G * max(cout, 1) * k * k * cin, device=dev)
    pix = N * H * W if tr else N * Ho * Wo
    fl = 2.0 * pix * cin * cout * k * k * G
    t_f = timeit(lambda: ops.conv_fwd(geom, x, wf, None, cp, y_pitch=op * G))
    t_d = timeit(lambda: ops.conv_dgrad(geom, dy, wb, tuple(x.shape), cp))
    t_w = timeit(lambda: ops.conv_wgrad(geom, x, dy, dw, cp, cin, cout, dw_gstride=cout * k * k * cin))
    print("%-32s %5.0f|%4.0f %5.0f|%4.0f %5.0f|%4.0f   %.1f GFLOP" % (name, t_f, fl / t_f / 1e6, t_d, fl / t_d / 1e6, t_w, fl / t_w / 1e6, fl / 1e9))

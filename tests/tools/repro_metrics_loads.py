"""Are the LOADS of the SSIM kernel wrong beside the slab weight-gradient kernel?  The diagnostics variant (div_mode 100) runs
the product kernel on images that hold their own element index, and logs every loaded value that is not the one its address
holds: {tap k, image a / b, lane, value found (= the address that was really read), expected, HW_ID, row, plane}.
    S2P_LIB=.../libs2p_hip_diag.so python tests/tools/repro_metrics_loads.py"""
import ctypes, os, sys, io, contextlib, collections, struct
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
sys.path.insert(1, os.path.join(R, "tools")); import uselib  # noqa: E402  (S2P_LIB=<second build> for an A/B)
import torch
from s2p_amd import _lib
from s2p_amd.models.networks.layers import ConvLayer
from s2p_amd.options.train_options import TrainOptions
from s2p_amd.models.pix2pix_model import Pix2PixModel
opt = TrainOptions().parse(["--env_type", "cheetah", "--batchSize", "64", "--precision", "bf16", "--gpu_ids", "0", "--checkpoints_dir", "/tmp/ck_ml"], quiet=True)
with contextlib.redirect_stdout(io.StringIO()):
    m = Pix2PixModel(opt)
L = m.netG.lay
g = torch.Generator().manual_seed(0)
bf = torch.bfloat16
a21 = torch.randn(64, 21, 21, 256, generator=g).to(bf).cuda(); b21 = torch.randn(64, 21, 21, 256, generator=g).to(bf).cuda()
N, C, H, W = 8, 3, 84, 84
img_a = torch.arange(N * C * H * W, dtype=torch.float32, device="cuda").view(N, C, H, W).contiguous()
img_b = (img_a + 0.5).contiguous()
lib = ctypes.CDLL(_lib._SO)
P = ctypes.c_void_p
lib.s2p_diag_image_metrics_map.argtypes = [P, P] + [ctypes.c_int] * 4 + [ctypes.c_float] + [P] * 3 + [ctypes.c_int, P]
side = torch.cuda.Stream()
slab = lambda: ConvLayer.wgrad_many([(L["b0c0"], a21, b21), (L["b0c1"], a21, b21)])      # noqa: E731
MAPN = N * C * (H - 10) * (W - 10)


def run():
    acc = torch.zeros(2, N, device="cuda")
    buf = torch.zeros(MAPN + 1 + 8 * 128 + 64, dtype=torch.float32, device="cuda")      # [log: count + 128 x 8 words | map]
    assert lib.s2p_diag_image_metrics_map(img_a.data_ptr(), img_b.data_ptr(), N, C, H, W, 2.0, acc[0].data_ptr(), acc[1].data_ptr(),
                                          buf.data_ptr(), 100, torch.cuda.current_stream().cuda_stream) == 0
    return acc, buf


def f32(u):
    return struct.unpack("<f", struct.pack("<I", u & 0xffffffff))[0]


def report(tag, buf):
    o = [v & 0xffffffff for v in buf[:1 + 8 * 128].view(torch.int32).cpu().tolist()]
    n = o[0]
    print("%s: %d loaded values differ from what their address holds" % (tag, n))
    ev = [o[1 + 8 * k: 9 + 8 * k] for k in range(min(n, 128))]
    if ev:
        print("   taps   :", sorted(collections.Counter(e[0] & 255 for e in ev).items()), " image b:", sum(1 for e in ev if e[0] & 256))
        print("   lanes  :", sorted(collections.Counter(e[1] for e in ev).items()))
        for e in ev[:24]:
            found, want = f32(e[2]), f32(e[3])
            print("   tap %2d %s lane %2d row %2d plane %2d tile %d: found %.1f expected %.1f (delta %+.1f elements = %+d rows %+d cols)  hw 0x%x" % (
                e[0] & 255, "b" if e[0] & 256 else "a", e[1], e[5], e[6], e[7], found, want, found - want,
                round((found - want) / W), int(found - want) - round((found - want) / W) * W, e[4]))


torch.cuda.synchronize()
acc, buf = run(); torch.cuda.synchronize()
report("quiet", buf)
for it in range(4):
    for _ in range(6):
        slab()
    with torch.cuda.stream(side):
        acc, buf = run()
    torch.cuda.synchronize()
    report("beside slab #%d" % it, buf)

"""PSNR / SSIM beside the slab weight-gradient kernel: WHICH numbers differ from the quiet run, through which path?
  (1) the product wrapper (s2p_amd.metrics.image_metrics: zeros -> kernel -> torch post-processing),
  (2) the C entry point called directly, raw per-image sums,
  (3) the diagnostics variant of the same kernel that also writes every per-position SSIM value (map diff: which lanes / rows).
    S2P_LIB=.../libs2p_hip_diag.so python tests/tools/repro_metrics_values.py"""
import ctypes, os, sys, io, contextlib, collections
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
sys.path.insert(1, os.path.join(R, "tools")); import uselib  # noqa: E402  (S2P_LIB=<second build> for an A/B)
import torch
from s2p_amd import _lib, metrics
from s2p_amd.models.networks.layers import ConvLayer
from s2p_amd.options.train_options import TrainOptions
from s2p_amd.models.pix2pix_model import Pix2PixModel

opt = TrainOptions().parse(["--env_type", "cheetah", "--batchSize", "64", "--precision", "bf16", "--gpu_ids", "0", "--checkpoints_dir", "/tmp/ck_mv"], quiet=True)
with contextlib.redirect_stdout(io.StringIO()):
    m = Pix2PixModel(opt)
L = m.netG.lay
g = torch.Generator().manual_seed(0)
bf = torch.bfloat16
a21 = torch.randn(64, 21, 21, 256, generator=g).to(bf).cuda(); b21 = torch.randn(64, 21, 21, 256, generator=g).to(bf).cuda()
N, C, H, W = 8, 3, 84, 84
img_a = (torch.rand(N, C, H, W, generator=g).cuda() * 2 - 1).contiguous()
img_b = (img_a + 0.1 * torch.randn(N, C, H, W, generator=g).cuda()).clamp(-1, 1).contiguous()
lib = ctypes.CDLL(_lib._SO)
P = ctypes.c_void_p
lib.s2p_image_metrics.argtypes = [P, P] + [ctypes.c_int] * 4 + [ctypes.c_float] + [P] * 3
lib.s2p_diag_image_metrics_map.argtypes = [P, P] + [ctypes.c_int] * 4 + [ctypes.c_float] + [P] * 3 + [ctypes.c_int, P]
side = torch.cuda.Stream()
slab = lambda: ConvLayer.wgrad_many([(L["b0c0"], a21, b21), (L["b0c1"], a21, b21)])      # noqa: E731


def wrapper():
    p, s = metrics.image_metrics(img_a, img_b)
    return torch.cat([p, s]).clone()


def raw():
    acc = torch.zeros(2, N, device="cuda")
    assert lib.s2p_image_metrics(img_a.data_ptr(), img_b.data_ptr(), N, C, H, W, 2.0, acc[0].data_ptr(), acc[1].data_ptr(),
                                 torch.cuda.current_stream().cuda_stream) == 0
    return acc.flatten().clone()


MAPS = []


DIV_MODE = 0


def raw_map():
    acc = torch.zeros(2, N, device="cuda")
    mp = torch.full((N * C, H - 10, W - 10), float("nan"), device="cuda")
    assert lib.s2p_diag_image_metrics_map(img_a.data_ptr(), img_b.data_ptr(), N, C, H, W, 2.0, acc[0].data_ptr(), acc[1].data_ptr(),
                                          mp.data_ptr(), DIV_MODE, torch.cuda.current_stream().cuda_stream) == 0
    MAPS.append(mp)
    return acc.flatten().clone()


def fmt(t):
    return " ".join("%.7g" % v for v in t.tolist())


MODES = {0: "compiler division (v_rcp, 1 wait state, use)", 4: "v_rcp + 2 wait states", 5: "v_rcp + 4 wait states", 1: "v_rcp + 16 wait states",
         2: "no transcendental instruction"}
for name, fn, mode in [("wrapper (psnr | ssim)", wrapper, 0), ("raw sums (sq | ssim)", raw, 0)] + [("raw sums + map, DIV %d: %s" % (k, v), raw_map, k) for k, v in MODES.items()]:
    MAPS.clear()
    DIV_MODE = mode
    torch.cuda.synchronize()
    q = fn(); torch.cuda.synchronize()
    q2 = fn(); torch.cuda.synchronize()
    with torch.cuda.stream(side):
        qs = fn()
    torch.cuda.synchronize()
    print("\n=== %s" % name)
    print(" quiet main     :", fmt(q))
    print(" quiet main #2  : max rel diff %.3e" % float(((q2 - q).abs() / q.abs()).max()))
    print(" quiet SIDE     : max rel diff %.3e" % float(((qs - q).abs() / q.abs()).max()))
    nq = len(MAPS)
    for it in range(4):
        for _ in range(6):
            slab()
        with torch.cuda.stream(side):
            o = fn()
        torch.cuda.synchronize()
        rel = (o - q).abs() / q.abs()
        print(" beside slab #%d : max rel diff first half %.3e second half %.3e" % (it, float(rel[:N].max()), float(rel[N:].max())))
        print("                 :", fmt(o))
    if MAPS:
        qm = MAPS[0]
        for k, mp in enumerate(MAPS[1:], 1):
            d = (mp != qm) & ~(torch.isnan(mp) & torch.isnan(qm))
            nb = int(d.sum())
            print(" map %d vs quiet map: %d of %d positions differ%s" % (k, nb, d.numel(), " (beside slab)" if k >= nq else ""))
            if nb:
                idx = d.nonzero()
                print("     planes:", sorted(collections.Counter(idx[:, 0].tolist()).items())[:24])
                print("     rows  :", sorted(collections.Counter(idx[:, 1].tolist()).items()))
                print("     cols  :", sorted(collections.Counter(idx[:, 2].tolist()).items()))
                for e in idx[:8].tolist():
                    print("     at %s quiet %.7g now %.7g" % (e, float(qm[tuple(e)]), float(mp[tuple(e)])))

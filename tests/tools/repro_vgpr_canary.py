"""Does anything write into the vector registers (or the EXEC mask) of a wave that shares a SIMD with the LDS-DMA conv kernels?
A canary wave (diagnostics build: s2p_diag_vgpr_canary) holds a pattern in 24 / 56 / 104 VGPRs and keeps re-checking it while ONE
kind of launch runs on another stream.
    S2P_LIB=.../libs2p_hip_diag.so python tests/tools/repro_vgpr_canary.py"""
import ctypes, os, sys, io, contextlib, collections
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
sys.path.insert(1, os.path.join(R, "tools")); import uselib  # noqa: E402  (S2P_LIB=<second build> for an A/B)
import torch
from s2p_amd import ops, _lib
from s2p_amd.models.networks.layers import ConvLayer
from s2p_amd.options.train_options import TrainOptions
from s2p_amd.models.pix2pix_model import Pix2PixModel
opt = TrainOptions().parse(["--env_type", "cheetah", "--batchSize", "64", "--precision", "bf16", "--gpu_ids", "0", "--checkpoints_dir", "/tmp/ck_vc"], quiet=True)
with contextlib.redirect_stdout(io.StringIO()):
    m = Pix2PixModel(opt)
L = m.netG.lay
g = torch.Generator().manual_seed(0)
bf = torch.bfloat16
a21 = torch.randn(64, 21, 21, 256, generator=g).to(bf).cuda(); b21 = torch.randn(64, 21, 21, 256, generator=g).to(bf).cuda()
a84 = torch.randn(64, 84, 84, 64, generator=g).to(bf).cuda(); d42 = torch.randn(64, 42, 42, 128, generator=g).to(bf).cuda()
lib = ctypes.CDLL(_lib._SO)
lib.s2p_diag_vgpr_canary.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
loads = {
    "none": lambda: None,
    "slab wgrad (ResBlk)": lambda: ConvLayer.wgrad_many([(L["b0c0"], a21, b21), (L["b0c1"], a21, b21)]),
    "conv_dma (down0 fwd, stride 2)": lambda: L["down0"].fwd(a84),
    "wgrad_dma (down0 wgrad)": lambda: L["down0"].wgrad(a84, d42),
    "plane-resident conv (ResBlk fwd)": lambda: L["b0c0"].fwd(a21),
}
side = torch.cuda.Stream()


def hw(h):
    return "se%d cu%d simd%d wave%d" % ((h >> 13) & 7, (h >> 8) & 0xf, (h >> 4) & 3, h & 0xf)


for regs in (24, 56, 104):
    for name, fn in loads.items():
        out = torch.zeros(1 + 8 * 64, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        for it in range(4):
            for _ in range(6):
                fn()
            with torch.cuda.stream(side):
                rc = lib.s2p_diag_vgpr_canary(2048, regs, 200, out.data_ptr(), side.cuda_stream)
                assert rc == 0
            torch.cuda.synchronize()
        o = [v & 0xffffffff for v in out.cpu().tolist()]
        n = o[0]
        print("canary %3d VGPRs beside %-34s: %6d register / EXEC events" % (regs, name, n), flush=True)
        ev = [o[1 + 8 * k: 9 + 8 * k] for k in range(min(n, 64))]
        if ev:
            print("    register indices:", sorted(collections.Counter(e[0] for e in ev).items()))
            print("    lanes           :", sorted(collections.Counter(e[1] for e in ev).items()))
            print("    GPR_ALLOC       :", collections.Counter(hex(e[4]) for e in ev).most_common(6))
            for e in ev[:12]:
                want = 0xC0DE0000 ^ (e[0] << 8) ^ e[1]
                print("    reg %3d lane %2d found 0x%08x (pattern 0x%08x) %s gpr_alloc 0x%x spin %d exec %08x_%08x" % (e[0], e[1], e[2], want, hw(e[3]), e[4], e[5], e[7], e[6]))

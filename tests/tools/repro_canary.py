"""Do the LDS-DMA kernels write outside their own LDS allocation?  An LDS canary workgroup (diagnostics build:
s2p_diag_lds_canary) holds a pattern in its LDS and keeps re-reading it while ONE kind of conv launch runs on another stream;
any word that changes was written by a co-resident workgroup of the other kernel.
    S2P_LIB=.../libs2p_hip_diag.so python tests/tools/repro_canary.py"""
import ctypes, os, sys, io, contextlib
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
sys.path.insert(1, os.path.join(R, "tools")); import uselib  # noqa: E402  (S2P_LIB=<second build> for an A/B)
import torch
from s2p_amd import ops, _lib
from s2p_amd.models.networks.layers import ConvLayer
from s2p_amd.options.train_options import TrainOptions
from s2p_amd.models.pix2pix_model import Pix2PixModel
opt = TrainOptions().parse(["--env_type", "cheetah", "--batchSize", "64", "--precision", "bf16", "--gpu_ids", "0", "--checkpoints_dir", "/tmp/ck_c"], quiet=True)
with contextlib.redirect_stdout(io.StringIO()):
    m = Pix2PixModel(opt)
L = m.netG.lay
g = torch.Generator().manual_seed(0)
bf = torch.bfloat16
a21 = torch.randn(64, 21, 21, 256, generator=g).to(bf).cuda(); b21 = torch.randn(64, 21, 21, 256, generator=g).to(bf).cuda()
a84 = torch.randn(64, 84, 84, 64, generator=g).to(bf).cuda(); d42 = torch.randn(64, 42, 42, 128, generator=g).to(bf).cuda()
actv = torch.randn(64, 21, 21, 1536, generator=g).to(bf).cuda()
v10 = torch.randn(128, 10, 10, 512, generator=g).to(bf).cuda()
lib = ctypes.CDLL(_lib._SO)
lib.s2p_diag_lds_canary.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
loads = {
    "none": lambda: None,
    "plane-resident conv (ResBlk fwd)": lambda: L["b0c0"].fwd(a21),
    "conv_dma (down0 fwd, stride 2)": lambda: L["down0"].fwd(a84),
    "conv_dma (down0 dgrad, phases)": lambda: L["down0"].dgrad(d42, a84.shape),
    "wgrad_dma (down0 wgrad)": lambda: L["down0"].wgrad(a84, d42),
    "slab wgrad (ResBlk)": lambda: ConvLayer.wgrad_many([(L["b0c0"], a21, b21), (L["b0c1"], a21, b21)]),
    "halo conv (512 -> 512 @ 10x10)": lambda: ops.conv_fwd(ops.ConvGeom(512, 512, 3, 1, 1), v10, torch.zeros(1, 512, 9, 512, dtype=bf, device="cuda"), None, 512),
}
side = torch.cuda.Stream()
for words in (4096, 8192):
    for name, fn in loads.items():
        out = torch.zeros(64, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        for it in range(6):
            for _ in range(6):
                fn()
            with torch.cuda.stream(side):
                rc = lib.s2p_diag_lds_canary(512, words, 40, out.data_ptr(), side.cuda_stream)
                assert rc == 0
            torch.cuda.synchronize()
        o = out.cpu().tolist()
        hits = [(o[1 + 2 * k], hex(o[2 + 2 * k] & 0xffffffff)) for k in range(min(o[0], 6))]
        print("canary %5d B beside %-36s: %6d words overwritten %s" % (words * 4, name, o[0], hits), flush=True)

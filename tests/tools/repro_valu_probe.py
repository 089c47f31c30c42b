"""Single-instruction VALU victims (diagnostics build: s2p_diag_valu_probe) beside the REAL conv kernels: which instruction form
loses results, in which lanes?
    S2P_LIB=.../libs2p_hip_diag.so python tests/tools/repro_valu_probe.py"""
import ctypes, os, sys, io, contextlib, collections
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
sys.path.insert(1, os.path.join(R, "tools")); import uselib  # noqa: E402  (S2P_LIB=<second build> for an A/B)
import torch
from s2p_amd import ops, _lib
from s2p_amd.models.networks.layers import ConvLayer
from s2p_amd.options.train_options import TrainOptions
from s2p_amd.models.pix2pix_model import Pix2PixModel
opt = TrainOptions().parse(["--env_type", "cheetah", "--batchSize", "64", "--precision", "bf16", "--gpu_ids", "0", "--checkpoints_dir", "/tmp/ck_vp"], quiet=True)
with contextlib.redirect_stdout(io.StringIO()):
    m = Pix2PixModel(opt)
L = m.netG.lay
g = torch.Generator().manual_seed(0)
bf = torch.bfloat16
a21 = torch.randn(64, 21, 21, 256, generator=g).to(bf).cuda(); b21 = torch.randn(64, 21, 21, 256, generator=g).to(bf).cuda()
a84 = torch.randn(64, 84, 84, 64, generator=g).to(bf).cuda(); d42 = torch.randn(64, 42, 42, 128, generator=g).to(bf).cuda()
lib = ctypes.CDLL(_lib._SO)
P = ctypes.c_void_p
lib.s2p_diag_valu_probe.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, P, P, P]
KINDS = ["v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32 (SGPR-pair src)", "v_pk_mov_b32 chain", "v_mov_b64 chain",
         "v_pk_fma_f16", "v_fma_f64", "v_pk_mul_f32 op_sel", "v_rcp_f32 + 1 wait state + use"]
loads = {
    "(quiet)": lambda: None,
    "slab wgrad (ResBlk)": lambda: ConvLayer.wgrad_many([(L["b0c0"], a21, b21), (L["b0c1"], a21, b21)]),
    "conv_dma (down0 fwd, stride 2)": lambda: L["down0"].fwd(a84),
    "wgrad_dma (down0 wgrad)": lambda: L["down0"].wgrad(a84, d42),
    "plane-resident conv (ResBlk fwd)": lambda: L["b0c0"].fwd(a21),
}
side = torch.cuda.Stream()
BLOCKS, ITERS = 4096, 4096
print("%-34s | %-32s | wrong lanes of %d x 3 runs, by 16-lane group [0-15 16-31 32-47 48-63] | example" % ("aggressor (main stream)", "victim (side stream)", BLOCKS * 64))
for name, fn in loads.items():
    for kind, kname in enumerate(KINDS):
        grp = [0, 0, 0, 0]; ex = ""
        for it in range(3):
            out = torch.full((BLOCKS * 64,), -1, dtype=torch.int32, device="cuda"); val = torch.zeros(BLOCKS * 64 * 2, device="cuda")
            torch.cuda.synchronize()
            for _ in range(6):
                fn()
            with torch.cuda.stream(side):
                assert lib.s2p_diag_valu_probe(kind, BLOCKS, ITERS, out.data_ptr(), val.data_ptr(), side.cuda_stream) == 0
            torch.cuda.synchronize()
            bad = (out != 0).nonzero().flatten()
            for q in range(4):
                grp[q] += int((((bad % 64) // 16) == q).sum())
            if len(bad) and not ex:
                b = int(bad[0]); ex = "lane %d: got %.9g / %.9g" % (b % 64, float(val[2 * b]), float(val[2 * b + 1]))
        print("%-34s | %-32s | %7d [%6d %6d %6d %6d] | %s" % (name, kname, sum(grp), *grp, ex), flush=True)

"""Does a kernel of the conv family disturb a small LDS-using kernel (the state path's linear kernels) that shares its CUs?
The split-K dgrad of the 256 -> 6144 affine layer runs on a side stream while ONE kind of conv launch runs on the main stream;
its result is compared bit for bit with the quiet result."""
import os, sys, io, contextlib
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch
from s2p_amd import ops
from s2p_amd.options.train_options import TrainOptions
from s2p_amd.models.pix2pix_model import Pix2PixModel
opt = TrainOptions().parse(["--env_type", "cheetah", "--batchSize", "64", "--precision", "bf16", "--gpu_ids", "0", "--checkpoints_dir", "/tmp/ck_lds"], quiet=True)
with contextlib.redirect_stdout(io.StringIO()):
    m = Pix2PixModel(opt)
L = m.netG.lay
M, K, N = 64, 256, 6144
g = torch.Generator().manual_seed(0)
x = torch.randn(M, K, generator=g).cuda(); dy = torch.randn(M, N, generator=g).cuda()
w_bwd = torch.randn(1, K, 1, N, generator=g).cuda().contiguous()
bf = torch.bfloat16
a21 = torch.randn(64, 21, 21, 256, generator=g).to(bf).cuda()
a84 = torch.randn(64, 84, 84, 64, generator=g).to(bf).cuda()
a42 = torch.randn(64, 42, 42, 128, generator=g).to(bf).cuda()
d42 = torch.randn(64, 42, 42, 128, generator=g).to(bf).cuda()
actv = torch.randn(64, 21, 21, 1536, generator=g).to(bf).cuda()
loads = {
    "none": lambda: None,
    "halo conv (ResBlk fwd)": lambda: L["b0c0"].fwd(a21),
    "halo conv (ResBlk dgrad)": lambda: L["b0c0"].dgrad(a21, a21.shape),
    "grouped halo conv (gamma/beta fwd)": lambda: L["gb"].fwd(actv),
    "conv_dma (down0 fwd, stride 2)": lambda: L["down0"].fwd(a84),
    "conv_dma (down0 dgrad, phases)": lambda: L["down0"].dgrad(d42, a84.shape),
    "wgrad_dma (down0 wgrad)": lambda: L["down0"].wgrad(a84, d42),
    "slab wgrad (ResBlk)": lambda: type(L["b0c0"]).wgrad_many([(L["b0c0"], a21, a21), (L["b0c1"], a21, a21)]),
    "IN fused fwd": lambda: ops.in_norm_fwd(a21, 256, act=1),
}
side = torch.cuda.Stream()
ref = None
ITERS = int(os.environ.get("REPRO_ITERS", "30"))
ONLY = os.environ.get("REPRO_ONLY", "")
for name, fn in loads.items():
    if ONLY and not any(k in name for k in ONLY.split(",")):
        continue
    bad = 0
    for it in range(ITERS):
        dw = torch.zeros(N * K, device="cuda"); db = torch.zeros(N, device="cuda")
        torch.cuda.synchronize()
        for _ in range(6):
            fn()
        with torch.cuda.stream(side):
            dx = ops.linear_bwd(x, dy, None, w_bwd, K, K, N, 0, 0.0, dw, db)
        torch.cuda.synchronize()
        if ref is None:
            ref = dx.clone()
        bad += int(not torch.equal(dx, ref))
    print("%-40s: %2d of %d results differ from the quiet result" % (name, bad, ITERS), flush=True)

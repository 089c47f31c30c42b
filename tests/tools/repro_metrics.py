"""image_metrics beside the slab weight-gradient kernel: how large is the difference to the quiet result?"""
import os, sys, io, contextlib
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
sys.path.insert(1, os.path.join(R, "tools")); import uselib  # noqa: E402  (S2P_LIB=<second build> for an A/B)
import torch
from s2p_amd import ops, metrics
from s2p_amd.models.networks.layers import ConvLayer
from s2p_amd.options.train_options import TrainOptions
from s2p_amd.models.pix2pix_model import Pix2PixModel
opt = TrainOptions().parse(["--env_type", "cheetah", "--batchSize", "64", "--precision", "bf16", "--gpu_ids", "0", "--checkpoints_dir", "/tmp/ck_m"], quiet=True)
with contextlib.redirect_stdout(io.StringIO()):
    m = Pix2PixModel(opt)
L = m.netG.lay
g = torch.Generator().manual_seed(0)
bf = torch.bfloat16
a21 = torch.randn(64, 21, 21, 256, generator=g).to(bf).cuda(); b21 = torch.randn(64, 21, 21, 256, generator=g).to(bf).cuda()
img_a = torch.rand(8, 3, 84, 84, generator=g).cuda() * 2 - 1
img_b = (img_a + 0.1 * torch.randn(8, 3, 84, 84, generator=g).cuda()).clamp(-1, 1)
side = torch.cuda.Stream()
torch.cuda.synchronize()
q = [torch.cat(metrics.image_metrics(img_a, img_b)).clone() for _ in range(4)]
torch.cuda.synchronize()
print("quiet runs differ among themselves by (max rel):", max(float(((x - q[0]).abs() / q[0].abs()).max()) for x in q))
for it in range(4):
    for _ in range(6):
        ConvLayer.wgrad_many([(L["b0c0"], a21, b21), (L["b0c1"], a21, b21)])
    with torch.cuda.stream(side):
        o = torch.cat(metrics.image_metrics(img_a, img_b)).clone()
    torch.cuda.synchronize()
    print("beside slab wgrad: max rel diff %.3e\n   psnr %s\n   vs   %s\n   ssim %s\n   vs   %s" % (float(((o - q[0]).abs() / q[0].abs()).max()), o[:8].tolist(), q[0][:8].tolist(), o[8:].tolist(), q[0][8:].tolist()))

import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "oracle")); sys.path.insert(0, os.path.join(R, "tests"))
sys.path.insert(1, os.path.join(R, "tools")); import uselib  # noqa: E402  (S2P_LIB=<second build> for an A/B)
import torch
import s2p_oracle as O
from s2p_amd import metrics
from s2p_amd.rollout import rollout
from s2p_amd.options.test_options import TestOptions
from s2p_amd.models.pix2pix_model import Pix2PixModel
from test_model_gpu import randomize
spec = O.Spec()
pg = randomize(O.init_params(O.generator_param_shapes(spec), 1), 11, 1.0)
ms = {}
for prec in ("fp32", "bf16"):
    opt = TestOptions().parse(["--env_type", "cheetah", "--gpu_ids", "0", "--random_init", "--precision", prec, "--checkpoints_dir", "/tmp/ck256", "--crop_size", "256"], quiet=True)
    m = Pix2PixModel(opt); m.netG.load_state_dict(pg); ms[prec] = m
g = torch.Generator().manual_seed(22)
for B in (2, 16):
    prev = torch.rand(B, 3, 256, 256, generator=g) * 2 - 1
    states = torch.randn(B, 2, 17, generator=g)
    with torch.no_grad():
        ya = ms["fp32"].netG(prev.cuda(), states[:, 0].cuda()); yb = ms["bf16"].netG(prev.cuda(), states[:, 0].cuda())
    ra = rollout(ms["fp32"].netG, prev, states); rb = rollout(ms["bf16"].netG, prev, states)
    torch.cuda.synchronize()
    p, s = metrics.image_metrics(yb, ya)
    p2, s2 = metrics.image_metrics(rb[:, 0], ra[:, 0])
    print("B=%d  forward(): bf16 vs fp32 max %.3e psnr min %.1f ssim min %.3f | rollout step 1: max %.3e psnr %.1f ssim %.3f | rollout vs forward fp32 max %.3e, bf16 max %.3e" % (
        B, float((ya - yb).abs().max()), float(p.min()), float(s.min()), float((ra[:, 0] - rb[:, 0]).abs().max()), float(p2.min()), float(s2.min()),
        float((ra[:, 0] - ya).abs().max()), float((rb[:, 0] - yb).abs().max())))

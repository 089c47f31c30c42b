"""Does the 256x256 generator forward depend on how the batch is split?  (InstanceNorm is per-sample: it must not.)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "oracle"))
sys.path.insert(1, os.path.join(R, "tools")); import uselib  # noqa: E402  (S2P_LIB=<second build> for an A/B)
import torch
import s2p_oracle as O
from s2p_amd.options.test_options import TestOptions
from s2p_amd.models.pix2pix_model import Pix2PixModel
spec = O.Spec()
pg = O.init_params(O.generator_param_shapes(spec), 1)
g = torch.Generator().manual_seed(11)
pg = {k: (v + 0.05 * torch.randn(v.shape, generator=g)) for k, v in pg.items()}
prev = (torch.rand(16, 3, 256, 256, generator=g) * 2 - 1).cuda()
state = torch.randn(16, 17, generator=g).cuda()
outs = {}
for prec in ("fp32", "bf16"):
    opt = TestOptions().parse(["--env_type", "cheetah", "--gpu_ids", "0", "--random_init", "--precision", prec, "--checkpoints_dir", "/tmp/ck256", "--crop_size", "256"], quiet=True)
    m = Pix2PixModel(opt); m.netG.load_state_dict(pg)
    with torch.no_grad():
        y16 = m.netG(prev, state)
        y4 = torch.cat([m.netG(prev[i:i + 4], state[i:i + 4]) for i in range(0, 16, 4)])
        y1 = torch.cat([m.netG(prev[i:i + 1], state[i:i + 1]) for i in range(0, 4)])
    torch.cuda.synchronize()
    print("%s: batch 16 vs 4 x 4: max |diff| %.3e ; per-sample max %s" % (prec, float((y16 - y4).abs().max()), [round(float((y16[i] - y4[i]).abs().max()), 4) for i in range(16)]))
    print("%s: batch 4 vs 4 x 1: max |diff| %.3e" % (prec, float((y4[:4] - y1).abs().max())))
    outs[prec] = (y16, y4)
print("bf16 vs fp32 (batch 16): max %.3e ; (4 x 4): max %.3e" % (float((outs["bf16"][0] - outs["fp32"][0]).abs().max()), float((outs["bf16"][1] - outs["fp32"][1]).abs().max())))

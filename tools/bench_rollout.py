"""BASELINE.json configs[4]: N-step rollout generation (seq_len 32), 256x256 frames, bs 16 per GPU, netG=s2p.
Reports frames/s for the eager rollout (s2p_amd.rollout.rollout) in bf16 and fp32."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
sys.path.insert(1, os.path.join(R, "tools")); import uselib  # noqa: E402  (S2P_LIB=<second build> for an A/B)
import io, contextlib
import torch
from s2p_amd.options.test_options import TestOptions
from s2p_amd.models.pix2pix_model import Pix2PixModel
from s2p_amd.rollout import rollout
B, T, S = 16, 32, 256
for prec in ("bf16", "fp32"):
    opt = TestOptions().parse(["--env_type", "cheetah", "--gpu_ids", "0", "--precision", prec, "--random_init",
                               "--crop_size", str(S), "--checkpoints_dir", "/tmp/ck_roll"], quiet=True)
    with contextlib.redirect_stdout(io.StringIO()):
        model = Pix2PixModel(opt)
    g = torch.Generator().manual_seed(0)
    img0 = (torch.rand(B, 3, S, S, generator=g) * 2 - 1).cuda()
    states = torch.randn(B, T, opt.state_dim, generator=g).cuda()
    rollout(model.netG, img0, states[:, :2]); torch.cuda.synchronize()
    t0 = time.time(); out = rollout(model.netG, img0, states); torch.cuda.synchronize(); dt = time.time() - t0
    assert out.shape == (B, T, 3, S, S) and bool(torch.isfinite(out).all())
    print("%s  rollout B=%d T=%d %dx%d: %.1f ms total, %.2f ms per generator step, %.0f frames/s"
          % (prec, B, T, S, S, dt * 1e3, dt / T * 1e3, B * T / dt), flush=True)

# BASELINE.json configs[1]: single-GPU generator forward, 84x84, bs 64 (fp32 is the parity configuration)
for prec in ("fp32", "bf16"):
    opt = TestOptions().parse(["--env_type", "cheetah", "--gpu_ids", "0", "--precision", prec, "--random_init",
                               "--crop_size", "84", "--checkpoints_dir", "/tmp/ck_roll"], quiet=True)
    with contextlib.redirect_stdout(io.StringIO()):
        model = Pix2PixModel(opt)
    g = torch.Generator().manual_seed(1)
    prev = (torch.rand(64, 3, 84, 84, generator=g) * 2 - 1).cuda(); st = torch.randn(64, opt.state_dim, generator=g).cuda()
    with torch.no_grad():
        for _ in range(3): model.netG(prev, st)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(20): model.netG(prev, st)
        torch.cuda.synchronize(); dt = (time.time() - t0) / 20
    print("%s  generator forward bs=64 84x84: %.2f ms, %.0f images/s (%.0f TFLOP/s at 13.83 GFLOP/img)"
          % (prec, dt * 1e3, 64 / dt, 13.83e9 * 64 / dt / 1e12), flush=True)

"""Adam + weight repack of both networks, us per call (hipGraph).  S2P_LIB=... to A/B two builds."""
import os, sys, io, contextlib
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
sys.path.insert(1, os.path.join(R, "tools")); import uselib  # noqa: E402  (S2P_LIB=<second build> for an A/B)
import torch
from s2p_amd.options.train_options import TrainOptions
from s2p_amd.trainers.pix2pix_trainer import Pix2PixTrainer
opt = TrainOptions().parse(["--env_type", "cheetah", "--batchSize", "4", "--precision", "bf16", "--gpu_ids", "0",
                            "--checkpoints_dir", "/tmp/ab_ck"], quiet=True)
with contextlib.redirect_stdout(io.StringIO()):
    tr = Pix2PixTrainer(opt)
for name, net in (("G", tr.pix2pix_model.netG), ("D", tr.pix2pix_model.netD)):
    st = net.store
    for _ in range(3): st.repack()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10): st.repack()
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    print("%s: repack %.1f us (%.1f M parameters)" % (name, e0.elapsed_time(e1) / 100 * 1e3, st.numel / 1e6))

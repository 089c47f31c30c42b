"""Which gradients of one G+D train step are bitwise reproducible?  Runs the step twice from the same weights (bf16 path) and
lists every parameter whose gradient differs between the runs (fp32 atomics somewhere on its weight-gradient path)."""
import os, sys, io, contextlib
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
sys.path.insert(1, os.path.join(R, "tools")); import uselib  # noqa: E402  (S2P_LIB=<second build> for an A/B)
import torch
from s2p_amd.options.train_options import TrainOptions
from s2p_amd.trainers.pix2pix_trainer import Pix2PixTrainer

B = int(os.environ.get("BATCH", "16"))
import importlib
for kv in filter(None, os.environ.get("FLAGS", "").split(",")):        # FLAGS=s2p_amd.models.networks.generator.COND_SIDE=False,...
    k, v = kv.split("="); mod, attr = k.rsplit(".", 1)
    setattr(importlib.import_module(mod), attr, eval(v))
from s2p_amd import ops as _ops
_ops.SERIALIZE = bool(int(os.environ.get("SERIAL", "0")))        # SERIAL=1: every launch on one stream (race or kernel?)
opt = TrainOptions().parse(["--env_type", "cheetah", "--batchSize", str(B), "--precision", "bf16", "--gpu_ids", "0",
                            "--checkpoints_dir", "/tmp/repro_ck"], quiet=True)
with contextlib.redirect_stdout(io.StringIO()):
    tr = Pix2PixTrainer(opt)
m = tr.pix2pix_model
g = torch.Generator().manual_seed(0)
data = dict(prev_image=(torch.rand(B, 3, 84, 84, generator=g) * 2 - 1).cuda(), image=(torch.rand(B, 3, 84, 84, generator=g) * 2 - 1).cuda(),
            state=torch.randn(B, 17, generator=g).cuda())
wG0, wD0 = m.netG.store.master.clone(), m.netD.store.master.clone()


def run():
    m.netG.store.master.copy_(wG0); m.netG.store.repack()
    m.netD.store.master.copy_(wD0); m.netD.store.repack()
    tr.optimizer_G.zero_grad()
    L, _ = m(data, mode="generator"); tr._backward(L)
    gG = {k: p.grad.detach().clone() for k, p in m.netG.named_parameters()}
    tr.optimizer_D.zero_grad()
    LD = m(data, mode="discriminator"); tr._backward(LD)
    gD = {k: p.grad.detach().clone() for k, p in m.netD.named_parameters()}
    torch.cuda.synchronize()
    return gG, gD, {k: float(v) for k, v in {**L, **LD}.items()}


a = run(); b = run()
bad = 0
for name, ga, gb in (("G", a[0], b[0]), ("D", a[1], b[1])):
    for k in ga:
        if not torch.equal(ga[k], gb[k]):
            bad += 1
            d = (ga[k].double() - gb[k].double()).norm() / (ga[k].double().norm() + 1e-30)
            print("%s %-44s differs: rel-L2 %.2e" % (name, k, float(d)))
print("losses run 1:", a[2]); print("losses run 2:", b[2])
print("%d parameter gradients differ between two identical runs" % bad)

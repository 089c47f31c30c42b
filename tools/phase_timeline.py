"""Where the train step's wall time goes WITHOUT a profiler (rocprofv3 serialises the submission of graph branches, so its overlapped
timeline misplaces them): HIP events recorded on whatever stream a phase runs on, around the generator / discriminator / VGG forward
and backward passes, the ResBlk chains and the optimizer steps, in an EAGER step at batch 64 (the GPU time of the eager step is
within a few percent of the graph replay).  Prints start / end offsets (us) of every phase from the start of the step, median of 5."""
import os, sys, io, contextlib, collections
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
sys.path.insert(1, os.path.join(R, "tools")); import uselib  # noqa: E402  (S2P_LIB=<second build> for an A/B)
import torch
from s2p_amd.options.train_options import TrainOptions
from s2p_amd.trainers.pix2pix_trainer import Pix2PixTrainer
from s2p_amd.models.networks.generator import S2PGenerator
from s2p_amd.models.networks.discriminator import MultiscaleDiscriminator
from s2p_amd.models.networks import loss as loss_mod
from s2p_amd.models.networks.layers import ConvLayer
from s2p_amd.params import ParamStore

opt = TrainOptions().parse(["--env_type", "cheetah", "--batchSize", "64", "--precision", "bf16", "--gpu_ids", "0", "--checkpoints_dir", "/tmp/ck_pt"], quiet=True)
with contextlib.redirect_stdout(io.StringIO()):
    tr = Pix2PixTrainer(opt)
g = torch.Generator().manual_seed(0)
data = dict(prev_image=(torch.rand(64, 3, 84, 84, generator=g) * 2 - 1).cuda(), image=(torch.rand(64, 3, 84, 84, generator=g) * 2 - 1).cuda(),
            state=torch.randn(64, 17, generator=g).cuda())
LOG = None


def wrap(cls, name, label):
    orig = getattr(cls, name)

    def f(self, *a, **k):
        if LOG is None:
            return orig(self, *a, **k)
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
        out = orig(self, *a, **k)
        e1 = torch.cuda.Event(enable_timing=True); e1.record()
        LOG.append((label(self) if callable(label) else label, e0, e1))
        return out
    setattr(cls, name, f)


wrap(S2PGenerator, "fwd_nhwc", "G forward")
wrap(S2PGenerator, "bwd_nhwc", "G backward")
wrap(MultiscaleDiscriminator, "fwd_nhwc", "D forward")
wrap(MultiscaleDiscriminator, "bwd_nhwc", "D backward")
wrap(ParamStore, "adam_step", lambda s: "Adam + repack (%d params)" % s.numel)
for nm in dir(loss_mod.VGG19):
    if nm in ("fwd_nhwc", "bwd_nhwc", "features_nhwc", "forward_nhwc", "backward_nhwc"):
        wrap(loss_mod.VGG19, nm, "VGG " + nm)
# the ResBlk chains: first / last fused launch of a forward and of a backward
of, od = ConvLayer.fwd_mat, ConvLayer.dgrad_mat


def fm(self, *a, **k):
    if LOG is not None and self.geom.k == 3:
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
        out = of(self, *a, **k)
        e1 = torch.cuda.Event(enable_timing=True); e1.record()
        LOG.append(("  ResBlk fused conv fwd", e0, e1))
        return out
    return of(self, *a, **k)


def dm(self, *a, **k):
    if LOG is not None and self.geom.k == 3:
        e0 = torch.cuda.Event(enable_timing=True); e0.record()
        out = od(self, *a, **k)
        e1 = torch.cuda.Event(enable_timing=True); e1.record()
        LOG.append(("  ResBlk fused dgrad", e0, e1))
        return out
    return od(self, *a, **k)


ConvLayer.fwd_mat, ConvLayer.dgrad_mat = fm, dm


def step():
    tr.run_generator_one_step(data); tr.run_discriminator_one_step(data)


for _ in range(3):
    step()
torch.cuda.synchronize()
runs = []
for it in range(5):
    LOG = []
    t0 = torch.cuda.Event(enable_timing=True); t0.record()
    step()
    t1 = torch.cuda.Event(enable_timing=True); t1.record()
    torch.cuda.synchronize()
    rows = [(lab, t0.elapsed_time(a) * 1e3, t0.elapsed_time(b) * 1e3) for lab, a, b in LOG]
    runs.append((t0.elapsed_time(t1) * 1e3, rows))
    LOG = None
runs.sort(key=lambda r: r[0])
tot, rows = runs[len(runs) // 2]
print("eager step %.0f us (median of 5; graph replay of the same step is the bench number)" % tot)
# collapse the fused-launch entries into chains
out, chain = [], None
for lab, a, b in rows:
    if lab.startswith("  ResBlk"):
        if chain and chain[0] == lab and a - chain[2] < 200:
            chain[2] = b; chain[3] += 1
        else:
            if chain: out.append(tuple(chain))
            chain = [lab, a, b, 1]
    else:
        out.append((lab, a, b, 0))
if chain: out.append(tuple(chain))
for lab, a, b, n in sorted(out, key=lambda r: r[1]):
    print("%9.0f -> %9.0f  (%6.0f us)  %s%s" % (a, b, b - a, lab, " x%d" % n if n else ""))

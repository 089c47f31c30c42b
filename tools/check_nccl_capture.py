"""De-risk for bench.py at N>1: hipGraph capture of the step segments with an RCCL process group alive (1-rank group on
one GPU; the collectives stay outside the captured segments exactly as in bench.py)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
sys.path.insert(1, os.path.join(R, "tools")); import uselib  # noqa: E402  (S2P_LIB=<second build> for an A/B)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
t = torch.ones(1 << 20, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()      # communicator + watchdog are up
from s2p_amd.options.train_options import TrainOptions
from s2p_amd.trainers.pix2pix_trainer import Pix2PixTrainer
import io, contextlib
opt = TrainOptions().parse(["--env_type", "cheetah", "--batchSize", os.environ.get("BATCH", "8"), "--precision", "bf16", "--gpu_ids", "0",
                            "--checkpoints_dir", "/tmp/ck_nccl"], quiet=True)
with contextlib.redirect_stdout(io.StringIO()):
    tr = Pix2PixTrainer(opt)
g = torch.Generator().manual_seed(0); B = int(os.environ.get("BATCH", "8"))
data = dict(prev_image=(torch.rand(B, 3, 84, 84, generator=g) * 2 - 1).cuda(), image=(torch.rand(B, 3, 84, 84, generator=g) * 2 - 1).cuda(),
            state=torch.randn(B, 17, generator=g).cuda())
model = tr.pix2pix_model
def g_part():
    tr.optimizer_G.zero_grad(); L, gen = model(data, mode="generator"); sum(L.values()).mean().backward(); tr.g_losses = L
def d_part():
    tr.optimizer_D.zero_grad(); L = model(data, mode="discriminator"); sum(L.values()).mean().backward(); tr.d_losses = L
ar = lambda store: dist.all_reduce(store.grad)
segs = [g_part, lambda: ar(model.netG.store), lambda: (tr.optimizer_G.step(), d_part()), lambda: ar(model.netD.store), tr.optimizer_D.step]
for _ in range(2):
    for s in segs: s()
torch.cuda.synchronize()
graphs = []; side = torch.cuda.Stream()
for i, s in enumerate(segs):
    if i in (1, 3): graphs.append(None); s(); continue
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=side): s()
    graphs.append(gr)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(20):
    for gr, s in zip(graphs, segs):
        s() if gr is None else gr.replay()
torch.cuda.synchronize()
print("captured %d graph segments with an RCCL group alive; 20 segmented steps: %.2f ms/step; losses finite: %s"
      % (sum(g is not None for g in graphs), (time.time() - t0) / 20 * 1e3,
         all(float(v) == float(v) for v in {**tr.g_losses, **tr.d_losses}.values())))
dist.destroy_process_group()

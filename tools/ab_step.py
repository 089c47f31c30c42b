"""Same-box A/B of Python-level switches on the full train step (box-to-box spread on this pool is +-3 %, so variants are
compared inside ONE process: capture A, time it, capture B, time it, alternating).  Usage:
    python tools/ab_step.py autograd_nodes.OVERLAP_VGG [rounds [valueA valueB]]      (values: Python literals, default True False)
    S2P_LIB=.../libs2p_hip_diag.so python tools/ab_step.py lib:0 3 0 1      (a run-time switch of the diagnostics library: s2p_diag_set(0, value))
    S2P_LIB=.../libs2p_hip_diag.so python tools/ab_step.py env:S2P_WGRAD_BLOCKS 3 None 256   (an environment variable that library reads)
    python tools/ab_step.py tr:pipeline_d 3 False True          (an attribute of the trainer)"""
import os, sys, time, importlib
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
sys.path.insert(1, os.path.join(R, "tools")); import uselib  # noqa: E402  (S2P_LIB=<second build> for an A/B)
import torch
from s2p_amd.options.train_options import TrainOptions
from s2p_amd.trainers.pix2pix_trainer import Pix2PixTrainer
from s2p_amd.stepgraph import StepGraph
import io, contextlib

target = sys.argv[1] if len(sys.argv) > 1 else "autograd_nodes.OVERLAP_VGG"
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
vals = (eval(sys.argv[3]), eval(sys.argv[4])) if len(sys.argv) > 4 else (True, False)
if target.startswith("lib:"):
    import ctypes
    from s2p_amd import _lib

    class _LibSwitch:                      # setattr(mod, attr, val) -> s2p_diag_set(key, val)
        def __setattr__(self, k, v):
            assert ctypes.CDLL(_lib._SO).s2p_diag_set(int(k[1:]), int(v)) == 0
    mod, attr = _LibSwitch(), "k" + target[4:]
elif target.startswith("tr:"):             # an attribute of the trainer object (e.g. tr:pipeline_d), bound once the trainer exists
    class _Tr:
        def __setattr__(self, k, v):
            setattr(tr, k, v)
    mod, attr = _Tr(), target[3:]
elif target.startswith("env:"):            # an environment variable the DIAGNOSTICS library reads at every call (None: unset)
    class _Env:
        def __setattr__(self, k, v):
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = str(v)
    mod, attr = _Env(), target[4:]
else:
    modname, attr = target.rsplit(".", 1)
    mod = importlib.import_module(modname if modname.startswith("s2p_amd") else "s2p_amd.models." + modname)
opt = TrainOptions().parse(["--env_type", "cheetah", "--batchSize", "64", "--precision", "bf16", "--gpu_ids", "0",
                            "--checkpoints_dir", "/tmp/ab_ck"], quiet=True)
with contextlib.redirect_stdout(io.StringIO()):
    tr = Pix2PixTrainer(opt)
g = torch.Generator().manual_seed(0)
data = dict(prev_image=(torch.rand(64, 3, 84, 84, generator=g) * 2 - 1).cuda(), image=(torch.rand(64, 3, 84, 84, generator=g) * 2 - 1).cuda(),
            state=torch.randn(64, 17, generator=g).cuda())


def step():
    tr.run_generator_one_step(data); tr.run_discriminator_one_step(data)


def measure(val, n=30):
    setattr(mod, attr, val)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    sg = StepGraph(); tr.seg = sg
    sg.capture(step)
    for _ in range(5):
        sg.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        sg.replay()
    torch.cuda.synchronize()
    tr.seg = None
    return (time.perf_counter() - t0) / n * 1e3


res = {v: [] for v in vals}
for r in range(rounds):
    for val in vals:
        res[val].append(measure(val))
        print("round %d  %s=%s  %.3f ms/step" % (r, target, val, res[val][-1]), flush=True)
for val in vals:
    v = sorted(res[val])
    print("%s=%s: median %.3f ms  min %.3f ms" % (target, val, v[len(v) // 2], v[0]))

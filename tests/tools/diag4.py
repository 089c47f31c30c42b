import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("", "tests", "oracle"): sys.path.insert(0, os.path.join(R, p))
import torch, torch.nn.functional as F
import s2p_oracle as O
from s2p_amd import ops
from s2p_amd.options.train_options import TrainOptions
from s2p_amd.models.pix2pix_model import Pix2PixModel
from test_model_gpu import randomize, make_inputs
from test_kernels_gpu import nchw
def rel(a, b): return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
opt = TrainOptions().parse(["--precision", "fp32", "--batchSize", "2", "--checkpoints_dir", "/tmp/ck"], quiet=True)
m = Pix2PixModel(opt); spec = O.Spec()
pg = randomize(O.init_params(O.generator_param_shapes(spec), 1), 11, 1.0)
m.netG.load_state_dict(pg)
prev, state, real = make_inputs(2, 84, 84, 17)
rec = {"add": [], "inb": [], "dg": []}
_add, _inb, _dg = ops.add, ops.in_bwd, ops.conv_dgrad
def add(a, b, out=None):
    r = _add(a, b, out); rec["add"].append(r.clone()); return r
def inb(*a, **k):
    r = _inb(*a, **k); rec["inb"].append((a[0].clone(), r[0].clone())); return r
ops.add, ops.in_bwd = add, inb
y = m.netG(prev.cuda(), state.cuda())
r = torch.randn(y.shape, generator=torch.Generator().manual_seed(5))
(y * r.cuda()).sum().backward()
# oracle with retained intermediates
p = {k: v.double() for k, v in pg.items()}
pr, st = prev.double(), state.double()
w = O.state_mapping(p, st, spec)
x = F.relu(O.instance_norm(F.conv2d(F.pad(pr, (3, 3, 3, 3), mode="reflect"), p["stem.weight"])))
for i in range(2): x = F.relu(O.instance_norm(F.conv2d(x, p[f"down{i}.weight"], stride=2, padding=1)))
x.requires_grad_(True)
xs, mids = [x], []
for b in range(6):
    xin = xs[-1]
    nA = F.leaky_relu(O.mat_norm(p, f"blocks.{b}.norm_0", xin, pr, w), 0.2); nA.retain_grad()
    c0 = F.conv2d(nA, p[f"blocks.{b}.conv_0.weight"], p[f"blocks.{b}.conv_0.bias"], padding=1); c0.retain_grad()
    nB = F.leaky_relu(O.mat_norm(p, f"blocks.{b}.norm_1", c0, pr, w), 0.2); nB.retain_grad()
    xn = xin + F.conv2d(nB, p[f"blocks.{b}.conv_1.weight"], p[f"blocks.{b}.conv_1.bias"], padding=1); xn.retain_grad()
    xs.append(xn); mids.append((nA, c0, nB))
x = xs[-1]
for i in range(2):
    x = F.relu(O.instance_norm(F.conv_transpose2d(x, p[f"up{i}.weight"], stride=2, padding=1, output_padding=1)))
out = torch.tanh(F.conv2d(F.pad(x, (3, 3, 3, 3), mode="reflect"), p["out.weight"], p["out.bias"]))
(out * r.double()).sum().backward()
# rec["add"][j] is dx at the INPUT of block 5-j
for j, t in enumerate(rec["add"]):
    b = 5 - j
    print("block", b, "d(input)", rel(nchw(t, 256), xs[b].grad))
# in_bwd calls in backward order: dec up1, up0, then per block (norm_1, norm_0) from block 5 down
calls = rec["inb"]
print("n in_bwd calls", len(calls))
idx = 2
for b in reversed(range(6)):
    nA, c0, nB = mids[b]
    da1, dx1 = calls[idx]; da0, dx0 = calls[idx + 1]; idx += 2
    print("block", b, "d_nB", rel(nchw(da1, 256), nB.grad), "d_c0", rel(nchw(dx1, 256), c0.grad), "d_nA", rel(nchw(da0, 256), nA.grad))
# where is the error of block-5 d_nB?
nA, c0, nB = mids[5]
da1, dx1 = calls[2]
e = (nchw(da1, 256).double() - nB.grad)
print("err by sample", e.pow(2).sum((1, 2, 3)).sqrt().tolist(), "ref", nB.grad.pow(2).sum((1, 2, 3)).sqrt().tolist())
pm = e.pow(2).sum((0, 1)).sqrt()
print("err per-pixel map (rows):"); print((pm / pm.max()).round(decimals=2)[:, :21])
ec = e.pow(2).sum((0, 2, 3)).sqrt()
print("err per-channel top:", torch.topk(ec, 5), "median", ec.median())
# grad at block-5 output
print("dx at block5 output:", rel(nchw(calls[1][1], 256) if False else nchw(calls[1][1], 128), xs[0].grad) if False else "")

import math, sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("", "tests", "oracle"): sys.path.insert(0, os.path.join(R, p))
import torch, torch.nn.functional as F
from s2p_amd import ops
from test_kernels_gpu import nhwc, nchw, pack_fwd, pack_bwd, CONV_CASES
dev = torch.device("cuda:0"); dt = torch.float32
def rel(a, b): return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))
for case in CONV_CASES:
    cin, cout, k, s, p, tr, refl, H, W, N = case
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, cin, H, W, generator=g); w = torch.randn((cin, cout, k, k) if tr else (cout, cin, k, k), generator=g) / math.sqrt(cin * k * k)
    b = torch.randn(cout, generator=g); op = 1 if tr else 0
    xr = x.double().requires_grad_(True); wr = w.double().requires_grad_(True)
    if tr: yr = F.conv_transpose2d(xr, wr, b.double(), stride=s, padding=p, output_padding=op)
    elif refl: yr = F.conv2d(F.pad(xr, (p, p, p, p), mode="reflect"), wr, b.double(), stride=s)
    else: yr = F.conv2d(xr, wr, b.double(), stride=s, padding=p)
    dy = torch.randn(yr.shape, generator=g); yr.backward(dy.double())
    cin_pad, cout_pad = ops.pad_to(cin, 4), ops.pad_to(cout, 4)
    geom = ops.ConvGeom(cin, cout, k, s, p, transposed=tr, reflect=refl, output_padding=op)
    w_std = w.permute(1, 0, 2, 3) if tr else w
    xd = nhwc(x, cin_pad, dt, dev)
    y = ops.conv_fwd(geom, xd, pack_fwd(w_std, cin_pad, dt, dev), b.to(dev), cin_pad)
    dyd = nhwc(dy, cout_pad, dt, dev)
    dx = ops.conv_dgrad(geom, dyd, pack_bwd(w_std, cin_pad, cout_pad, dt, dev), tuple(xd.shape), cin_pad)
    rows, cols = (cin, cout) if tr else (cout, cin)
    dw = torch.zeros(rows, k * k, cols, device=dev)
    ops.conv_wgrad(geom, xd, dyd, dw, cin_pad, cin, cout)
    print(case, "fwd %.2e dgrad %.2e wgrad %.2e" % (rel(nchw(y, cout), yr.detach()), rel(nchw(dx, cin), xr.grad),
          rel(dw.cpu(), wr.grad.permute(0, 2, 3, 1).reshape(rows, k * k, cols))))

# layer-by-layer generator forward vs fp64 oracle
import s2p_oracle as O
from s2p_amd.options.train_options import TrainOptions
from s2p_amd.models.pix2pix_model import Pix2PixModel
from test_model_gpu import randomize, make_inputs
opt = TrainOptions().parse(["--precision", "fp32", "--batchSize", "2", "--checkpoints_dir", "/tmp/ck"], quiet=True)
m = Pix2PixModel(opt); spec = O.Spec()
pg = randomize(O.init_params(O.generator_param_shapes(spec), 1), 11, 1.0)
m.netG.load_state_dict(pg)
prev, state, real = make_inputs(2, 84, 84, 17)
img = ops.nchw_to_nhwc(prev.cuda(), dt, 4)
out, ctx = m.netG.fwd_nhwc(img, state.cuda())
p64 = {k: v.double() for k, v in pg.items()}
pr, st = prev.double(), state.double()
w = O.state_mapping(p64, st, spec)
print("w", rel(ctx["hs"][-1].view(2, -1).cpu(), w))
x = F.conv2d(F.pad(pr, (3, 3, 3, 3), mode="reflect"), p64["stem.weight"])
print("stem conv", rel(nchw(ctx["enc"][0][1], 64), x))
x = F.relu(O.instance_norm(x)); print("stem in", rel(nchw(ctx["enc"][0][3], 64), x))
for i in range(2):
    x = F.conv2d(x, p64[f"down{i}.weight"], stride=2, padding=1); print("down conv", i, rel(nchw(ctx["enc"][i + 1][1], x.shape[1]), x))
    x = F.relu(O.instance_norm(x)); print("down in", i, rel(nchw(ctx["enc"][i + 1][3], x.shape[1]), x))
for b in range(6):
    xb, sA, nA, c0, sB, nB = ctx["blocks"][b]
    print("block", b, "in", rel(nchw(xb, 256), x), end=" ")
    nAr = F.leaky_relu(O.mat_norm(p64, f"blocks.{b}.norm_0", x, pr, w), 0.2); print("nA", rel(nchw(nA, 256), nAr), end=" ")
    c0r = F.conv2d(nAr, p64[f"blocks.{b}.conv_0.weight"], p64[f"blocks.{b}.conv_0.bias"], padding=1); print("c0", rel(nchw(c0, 256), c0r), end=" ")
    nBr = F.leaky_relu(O.mat_norm(p64, f"blocks.{b}.norm_1", c0r, pr, w), 0.2); print("nB", rel(nchw(nB, 256), nBr))
    x = x + F.conv2d(nBr, p64[f"blocks.{b}.conv_1.weight"], p64[f"blocks.{b}.conv_1.bias"], padding=1)
print("final", rel(nchw(out, 3), O.generator_forward(p64, pr, st, spec)))

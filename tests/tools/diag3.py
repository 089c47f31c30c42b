import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("", "tests", "oracle"): sys.path.insert(0, os.path.join(R, p))
import torch
import s2p_oracle as O
from s2p_amd.options.train_options import TrainOptions
from s2p_amd.models.pix2pix_model import Pix2PixModel
from test_model_gpu import randomize, make_inputs, grad_errors, to64
opt = TrainOptions().parse(["--precision", "fp32", "--batchSize", "2", "--checkpoints_dir", "/tmp/ck"], quiet=True)
m = Pix2PixModel(opt); spec = O.Spec()
pg = randomize(O.init_params(O.generator_param_shapes(spec), 1), 11, 1.0)
m.netG.load_state_dict(pg)
prev, state, real = make_inputs(2, 84, 84, 17)
y = m.netG(prev.cuda(), state.cuda())
r = torch.randn(y.shape, generator=torch.Generator().manual_seed(5))
m.netG.store.zero_grad()
(y * r.cuda()).sum().backward()
pg64 = to64(pg)
(O.generator_forward(pg64, prev.double(), state.double(), spec) * r.double()).sum().backward()
errs = grad_errors(dict(m.netG.named_parameters()), pg64)
for k, (e, _) in errs.items():
    if "blocks." in k and not k.startswith("blocks.0.") and not k.startswith("blocks.5."): continue
    a = dict(m.netG.named_parameters())[k].grad.detach().cpu().double().flatten(); b = pg64[k].grad.flatten()
    print("%-40s relL2 %.2e  |hip| %.3e |ref| %.3e  cos %.8f" % (k, e, a.norm(), b.norm(), float((a @ b) / (a.norm() * b.norm() + 1e-30))))

"""bf16 / fp32 gradient error of the SLAC stacks vs the reference fixture, per parameter (diagnostic)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "oracle"))
sys.path.insert(1, os.path.join(R, "tools")); import uselib  # noqa: E402  (S2P_LIB=<second build> for an A/B)
import numpy as np, torch
import slac_oracle as SO
from s2p_amd.slac import Decoder, Encoder
GB = np.load(os.path.join(R, "tests", "golden", "slac_bwd_golden_v1.npz"))
pe, pd = (SO.make_params(SO.ENCODER_100, int(GB["seeds"][0])), SO.make_params(SO.DECODER_100, int(GB["seeds"][1])))
x, z, r_feat, r_img = SO.backward_case(int(GB["seeds"][2]), int(GB["seeds"][3]))
# fp32 CPU oracle full grads for an L2 comparison over ALL elements
for p in list(pe.values()) + list(pd.values()): p.requires_grad_(True)
zo = z.clone().requires_grad_(True)
(SO.encoder_forward(pe, x) * r_feat).sum().backward(); (SO.decoder_forward(pd, zo) * r_img).sum().backward()
for dtype in (torch.float32, torch.bfloat16):
    enc = Encoder(3, 256, 100, dtype=dtype).load_state_dict({k: v.detach() for k, v in pe.items()})
    dec = Decoder(288, 3, 1.0, 100, dtype=dtype).load_state_dict({k: v.detach() for k, v in pd.items()})
    zc = z.cuda().requires_grad_(True)
    (enc(x) * r_feat.cuda()).sum().backward(); img, _ = dec(zc); (img * r_img.cuda()).sum().backward()
    rel = lambda a, b: float((a.double().cpu() - b.double()).norm() / b.double().norm())
    print(dtype, "dz", "%.4f" % rel(zc.grad, zo.grad))
    for name, mod, ref in (("enc", enc, pe), ("dec", dec, pd)):
        for k, v in mod.state_dict(keep_vars=True).items():
            print("   %s.%-14s %.4f" % (name, k, rel(v.grad, ref[k].grad)))

"""N3 measurement: SLAC encoder / decoder forward on HIP (bf16) vs the CPU oracle restatement (16 host threads)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ("", "oracle"): sys.path.insert(0, os.path.join(R, p))
import torch
import slac_oracle as SO
from s2p_amd.slac import Encoder, Decoder
torch.set_num_threads(min(16, os.cpu_count() or 1))
pe, pd = SO.make_params(SO.ENCODER_100, 1), SO.make_params(SO.DECODER_100, 2)
def flops(spec, size_in):
    f, h = 0.0, size_in
    for kind, cin, cout, k, s, pad, op in spec:
        ho = (h + 2 * pad - k) // s + 1 if kind == "conv" else (h - 1) * s - 2 * pad + k + op
        f += 2.0 * cin * cout * k * k * (ho * ho if kind == "conv" else h * h)
        h = ho
    return f
fe, fd = flops(SO.ENCODER_100, 100), flops(SO.DECODER_100, 1)
B, S = 32, 9          # SLAC batch 32 sequences x (num_sequences + 1) frames
x = torch.rand(B, S, 3, 100, 100); z = torch.randn(B, S, 288)
for dt in (torch.bfloat16, torch.float32):
    enc, dec = Encoder(dtype=dt).load_state_dict(pe), Decoder(dtype=dt).load_state_dict(pd)
    xd, zd = x.cuda(), z.cuda()
    for _ in range(3): enc(xd); dec(zd)
    torch.cuda.synchronize(); n = 20
    t = time.time()
    for _ in range(n): enc(xd)
    torch.cuda.synchronize(); te = (time.time() - t) / n
    t = time.time()
    for _ in range(n): dec(zd)
    torch.cuda.synchronize(); td = (time.time() - t) / n
    print("%s  encoder %.3f ms (%.0f frames/s, %.1f TFLOP/s)   decoder %.3f ms (%.0f frames/s, %.1f TFLOP/s)"
          % (str(dt).split(".")[-1], te * 1e3, B * S / te, fe * B * S / te / 1e12, td * 1e3, B * S / td, fd * B * S / td / 1e12))
with torch.no_grad():
    t = time.time(); SO.encoder_forward(pe, x); te = time.time() - t
    t = time.time(); SO.decoder_forward(pd, z); td = time.time() - t
print("CPU oracle (fp32, %d threads): encoder %.0f frames/s, decoder %.0f frames/s" % (torch.get_num_threads(), B * S / te, B * S / td))

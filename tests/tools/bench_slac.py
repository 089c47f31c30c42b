"""N3 measurement: SLAC encoder / decoder forward and forward+backward on HIP vs the CPU oracle restatement (16 host threads)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
    # training pass: forward + backward of both stacks (parameter grads; latent grad for the decoder)
    zg = zd.clone().requires_grad_(True)
    r1 = torch.randn(B, S, 256, device="cuda"); r2 = torch.randn(B, S, 3, 100, 100, device="cuda")
    def train_pass():
        for p in list(enc.parameters()) + list(dec.parameters()): p.grad = None
        zg.grad = None
        (enc(xd) * r1).sum().backward()
        img, _ = dec(zg)
        (img * r2).sum().backward()
    for _ in range(3): train_pass()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n): train_pass()
    torch.cuda.synchronize(); tt = (time.time() - t) / n
    print("%s  encoder+decoder fwd+bwd %.3f ms (%.0f frames/s, %.1f TFLOP/s algorithmic at 3x fwd)"
          % (str(dt).split(".")[-1], tt * 1e3, B * S / tt, 3 * (fe + fd) * B * S / tt / 1e12))
with torch.no_grad():
    t = time.time(); SO.encoder_forward(pe, x); te = time.time() - t
    t = time.time(); SO.decoder_forward(pd, z); td = time.time() - t
print("CPU oracle (fp32, %d threads): encoder %.0f frames/s, decoder %.0f frames/s" % (torch.get_num_threads(), B * S / te, B * S / td))
for q in list(pe.values()) + list(pd.values()): q.requires_grad_(True)
zc = z.clone().requires_grad_(True)
t = time.time()
(SO.encoder_forward(pe, x) * torch.randn(B, S, 256)).sum().backward()
(SO.decoder_forward(pd, zc) * torch.randn(B, S, 3, 100, 100)).sum().backward()
print("CPU oracle fwd+bwd: %.0f frames/s" % (B * S / (time.time() - t)))


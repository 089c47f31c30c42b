"""N2 measurement: ensemble dynamics forward + post-processing, HIP vs the CPU oracle restatement (16 host threads)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ("", "oracle"): sys.path.insert(0, os.path.join(R, p))
import torch
import ensemble_oracle as EO
from s2p_amd.dynamics import EnsembleTransition
torch.set_num_threads(min(16, os.cpu_count() or 1))
g = torch.Generator().manual_seed(0)
E, H = 7, 256
sd = {}
for i, (a, b) in enumerate([(23, H), (H, H), (H, H)]):
    sd[f"backbones.{i}.weight"] = torch.randn(E, a, b, generator=g) / (2 * a ** 0.5); sd[f"backbones.{i}.bias"] = torch.zeros(E, 1, b)
sd["output_layer.weight"] = torch.randn(E, H, 36, generator=g) / 32; sd["output_layer.bias"] = torch.zeros(E, 1, 36)
sd["max_logstd"] = torch.ones(18); sd["min_logstd"] = -5 * torch.ones(18)
m = EnsembleTransition(17, 6, H, 3, E).load_state_dict(sd)
flop_per_sample = 2.0 * E * (23 * H + 2 * H * H + H * 36)
for B in (1000, 50000):
    x = torch.randn(B, 23, generator=g); idx = torch.randint(0, E, (B,), generator=g)
    xd = x.cuda()
    om, os_ = torch.zeros(17), torch.ones(17)
    for _ in range(3): m.rollout_step(xd, idx, om, os_, 0.0, 1.0)
    torch.cuda.synchronize(); t = time.time(); n = 20
    for _ in range(n): m.rollout_step(xd, idx, om, os_, 0.0, 1.0)
    torch.cuda.synchronize(); dt = (time.time() - t) / n
    t = time.time()
    with torch.no_grad():
        mean, std = EO.ensemble_forward(sd, x, 17); EO.rollout_postprocess(mean, std, idx, om, os_, 0.0, 1.0)
    dc = time.time() - t
    print("B=%6d  HIP %.3f ms (%.2f M samples/s, %.1f TFLOP/s fp32; fp32 MFMA peak 157)   CPU oracle %.1f ms (%.2f M samples/s)"
          % (B, dt * 1e3, B / dt / 1e6, flop_per_sample * B / dt / 1e12, dc * 1e3, B / dc / 1e6))

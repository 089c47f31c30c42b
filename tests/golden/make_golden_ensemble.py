"""Generates tests/golden/ensemble_golden_v1.npz by RUNNING THE REAL REFERENCE (`/root/reference/gaussian_ensemble.py`,
importable in the build container only).  The fixture holds data only: a small seeded reference model's state_dict
(hidden 64), inputs, and the reference's outputs; the post-processing of state_transition_rollout.py:192-204 (that
script itself needs h5py/dmc2gym and cannot be imported) is evaluated with the same torch expressions on the
reference model's own distribution.  Run:  python tests/golden/make_golden_ensemble.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
from gaussian_ensemble import EnsembleTransition  # noqa: E402  (the real reference)


def main():
    torch.manual_seed(20261003)
    obs_dim, act_dim, E = 17, 6, 7
    model = EnsembleTransition(obs_dim, act_dim, 64, 3, ensemble_size=E)
    with torch.no_grad():       # non-trivial biases / clamps so every term is exercised
        for n, p in model.named_parameters():
            if n.endswith("bias") and "saved" not in n:
                p.copy_(torch.randn_like(p) * 0.1)
        model.max_logstd.copy_(torch.rand(obs_dim + 1) * 1.5 - 0.5)
        model.min_logstd.copy_(-torch.rand(obs_dim + 1) * 3 - 2)
    B = 37
    x = torch.randn(B, obs_dim + act_dim)
    x[:, obs_dim:] = torch.rand(B, act_dim) * 2 - 1
    x[0] *= 30.0                 # drive some logstd values into both soft-clamp regimes
    with torch.no_grad():
        dist = model(x)
        mean, std = dist.mean, dist.stddev
        idx = torch.from_numpy(np.random.default_rng(5).integers(0, E, B))
        nom, nos = torch.randn(obs_dim), torch.rand(obs_dim) + 0.5
        rm, rs = 2.991, 1.092        # reward stats of world_model/.../normalize_configs_dict.pkl
        bi = torch.arange(B)
        next_obs = mean[:, :, :obs_dim][idx, bi] * nos + nom
        reward = mean[:, :, -1][idx, bi] * rs + rm
        modes = mean[:, :, :-1]
        dis = torch.max(torch.norm(modes - torch.mean(modes, dim=0), dim=-1, keepdim=True), dim=0)[0]
        ale = torch.max(torch.norm(std, dim=-1, keepdim=True), dim=0)[0]
    out = {"sd." + k: v.detach().numpy() for k, v in model.state_dict().items() if "saved" not in k}
    out.update(x=x.numpy(), mean=mean.numpy(), std=std.numpy(), idx=idx.numpy().astype(np.int32), next_obs_mean=nom.numpy(),
               next_obs_std=nos.numpy(), reward_stats=np.array([rm, rs], np.float32), next_obs=next_obs.numpy(),
               reward=reward.numpy(), disagreement=dis.numpy(), aleatoric=ale.numpy())
    path = os.path.join(HERE, "ensemble_golden_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

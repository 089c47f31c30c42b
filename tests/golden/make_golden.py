"""Generates tests/golden/s2p_golden_v1.npz from the CPU oracle (oracle/s2p_oracle.py).

PARITY UNPINNED: the reference checkout has no generator code, tests or golden vectors (SURVEY.md sections 0, 4), so
these vectors pin the build's own frozen spec (SPEC.md): they guard the oracle against regressions and give the HIP
path a committed expected output.  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import s2p_oracle as O  # noqa: E402


def golden_params(spec):
    def rnd(params, seed, gain):
        g = torch.Generator().manual_seed(seed)
        out = {}
        for k, v in params.items():
            if k.endswith(".bias"):
                out[k] = torch.randn(v.shape, generator=g) * 0.1
            else:
                out[k] = torch.randn(v.shape, generator=g) * gain / v[0].numel() ** 0.5
        return out
    pg = rnd(O.init_params(O.generator_param_shapes(spec), 1), 11, 1.0)
    pd = rnd(O.init_params(O.discriminator_param_shapes(spec), 2), 12, 1.0)
    pv = O.init_params(O.vgg_param_shapes(), 3, kaiming=True)
    return pg, pd, pv


def golden_inputs(N=1, H=84, W=84, S=17, seed=2026):
    g = torch.Generator().manual_seed(seed)
    prev = torch.rand(N, 3, H, W, generator=g) * 2 - 1
    real = torch.rand(N, 3, H, W, generator=g) * 2 - 1
    state = torch.randn(N, S, generator=g)
    return prev, state, real


def main():
    torch.set_num_threads(4)
    spec = O.Spec()
    pg, pd, pv = golden_params(spec)
    prev, state, real = golden_inputs()
    out = {}
    with torch.no_grad():
        fake = O.generator_forward(pg, prev, state, spec)
        w = O.state_mapping(pg, state, spec)
        pf, pr = O.discriminate(pd, prev, fake, real, spec)
        L, _ = O.generator_losses(pg, pd, pv, prev, state, real, spec)
        D = O.discriminator_losses(pg, pd, prev, state, real, spec)
        frames = O.rollout(pg, prev, torch.stack([state, state * 0.5, -state], 1), spec)
    out.update(prev=prev.numpy(), state=state.numpy(), real=real.numpy(), fake=fake.numpy(), w=w.numpy(),
               d_logits0=pf[0][-1].numpy(), d_logits1=pf[1][-1].numpy(),
               d_feat00=pf[0][0].numpy()[:, :4], rollout_last=frames[:, -1].numpy(),
               g_losses=np.array([float(L[k]) for k in ("GAN", "GAN_Feat", "VGG", "L1")], np.float64),
               d_losses=np.array([float(D[k]) for k in ("D_Fake", "D_real")], np.float64),
               param_checksum=np.array([float(sum(v.double().abs().sum() for v in p.values())) for p in (pg, pd, pv)]))
    path = os.path.join(HERE, "s2p_golden_v1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

"""Ensemble state-dynamics forward: the step that produces the predicted next states the
generator consumes.  Mirrors the reference interface `EnsembleTransition(obs_dim, action_dim, hidden_features,
hidden_layers, ensemble_size)` / `model(obs_action) -> Normal(mean, std)` (gaussian_ensemble.py:60-96) and the
per-batch post-processing of state_transition_rollout.py:180-204, running on HIP kernels:

  * the E members are ONE grouped 1x1 "conv" per layer on the exact-fp32 MFMA path (groups = E, the first layer reads
    the same input for every group via x_gstride = 0), Swish fused in the epilogue;
  * one fused head kernel does soft-clamp + exp, the 'local'-mode residual, member pick + de-normalisation and the
    disagreement / aleatoric reductions (s2p_ensemble_head).
Inference only (the reference rollout script never trains it).  No CPU fallback.
"""
import ctypes

import torch

from . import ops
from ._lib import ACT_NONE, ACT_SWISH, check, lib, ptr, stream
from .ops import ConvGeom, pad_to


class EnsembleTransition:
    def __init__(self, obs_dim, action_dim, hidden_features, hidden_layers, ensemble_size=7, device="cuda:0"):
        self.obs_dim, self.action_dim, self.hidden, self.n_hidden, self.E = obs_dim, action_dim, hidden_features, hidden_layers, ensemble_size
        self.D = obs_dim + 1
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("EnsembleTransition (HIP) needs a HIP device: there is no CPU fallback")
        self.layers = None

    def load_state_dict(self, sd):
        """Reference layout: weight [E, in, out], bias [E, 1, out] (gaussian_ensemble.py:27-28); saved_* ignored."""
        E, dev = self.E, self.device
        self.layers = []
        names = [f"backbones.{i}" for i in range(self.n_hidden)] + ["output_layer"]
        for li, name in enumerate(names):
            w = torch.as_tensor(sd[name + ".weight"], dtype=torch.float32)
            b = torch.as_tensor(sd[name + ".bias"], dtype=torch.float32)
            assert w.shape[0] == E, "ensemble size mismatch"
            cin, cout = w.shape[1], w.shape[2]
            cin_pad = pad_to(cin, 4)
            wp = torch.zeros(E, cout, 1, cin_pad)
            wp[:, :, 0, :cin] = w.permute(0, 2, 1)                       # packed [group][Cout][tap][Cin_pad]
            geom = ConvGeom(cin, cout, 1, groups=E, x_gstride=0 if li == 0 else cin, y_gstride=cout)
            self.layers.append((geom, wp.to(dev).contiguous(), b.reshape(E * cout).to(dev).contiguous(), cin_pad))
        self.min_logstd = torch.as_tensor(sd["min_logstd"], dtype=torch.float32).to(dev).contiguous()
        self.max_logstd = torch.as_tensor(sd["max_logstd"], dtype=torch.float32).to(dev).contiguous()
        assert self.layers[-1][0].cout == 2 * self.D
        return self

    def _raw(self, obs_action):
        B = obs_action.shape[0]
        xin_pitch = self.layers[0][3]
        x = torch.zeros((B, 1, 1, xin_pitch), dtype=torch.float32, device=self.device)
        x[:, 0, 0, :obs_action.shape[1]] = obs_action.to(self.device, torch.float32)
        h = x
        for li, (geom, w, b, cin_pad) in enumerate(self.layers):
            last = li == len(self.layers) - 1
            h = ops.conv_fwd(geom, h, w, b, cin_pad, y_pitch=self.E * geom.cout, act=ACT_NONE if last else ACT_SWISH)
        return x, h

    @torch.no_grad()
    def forward(self, obs_action):
        """obs_action: fp32 [B, obs_dim+action_dim] (normalised).  Returns (mean [E,B,D], std [E,B,D])."""
        x, raw = self._raw(obs_action)
        B = x.shape[0]
        mean = torch.empty((self.E, B, self.D), dtype=torch.float32, device=self.device)
        std = torch.empty_like(mean)
        check(lib().s2p_ensemble_head(ptr(raw), raw.shape[3], ptr(x), x.shape[3], B, self.E, self.D, ptr(self.min_logstd),
                                      ptr(self.max_logstd), ptr(mean), ptr(std), None, None, None, 0.0, 1.0, None, None,
                                      None, None, stream()), "s2p_ensemble_head")
        return mean, std

    __call__ = forward

    @torch.no_grad()
    def rollout_step(self, obs_action, ensemble_idx, next_obs_mean, next_obs_std, reward_mean, reward_std):
        """One trajectory batch of state_transition_rollout.py:180-204: returns (next_obs [B,obs_dim], reward [B],
        disagreement [B,1], aleatoric [B,1]) -- de-normalised prediction of the picked member + uncertainties."""
        x, raw = self._raw(obs_action)
        B, dev = x.shape[0], self.device
        idx = torch.as_tensor(ensemble_idx).to(dev, torch.int32).contiguous()
        om = torch.as_tensor(next_obs_mean, dtype=torch.float32).to(dev).contiguous()
        os_ = torch.as_tensor(next_obs_std, dtype=torch.float32).to(dev).contiguous()
        nobs = torch.empty((B, self.obs_dim), dtype=torch.float32, device=dev)
        rew = torch.empty((B,), dtype=torch.float32, device=dev)
        dis = torch.empty((B, 1), dtype=torch.float32, device=dev)
        ale = torch.empty((B, 1), dtype=torch.float32, device=dev)
        check(lib().s2p_ensemble_head(ptr(raw), raw.shape[3], ptr(x), x.shape[3], B, self.E, self.D, ptr(self.min_logstd),
                                      ptr(self.max_logstd), None, None, ptr(idx), ptr(om), ptr(os_), float(reward_mean),
                                      float(reward_std), ptr(nobs), ptr(rew), ptr(dis), ptr(ale), stream()), "s2p_ensemble_head")
        return nobs, rew, dis, ale

"""BaseNetwork: SPADE-lineage network base class (plugin contract of `--netG/--netD`):
`modify_commandline_options(parser, is_train)` static hook, `cls(opt)` constructor, `init_weights(init_type,
init_variance)`, `print_network()`.  Adds the MI355X flat parameter store hooks."""
import math

import torch
import torch.nn as nn

from ...params import ParamStore


class BaseNetwork(nn.Module):
    def __init__(self):
        super().__init__()
        self.store = ParamStore()
        self.compute_dtype = torch.float32
        self.finalized = False

    @staticmethod
    def modify_commandline_options(parser, is_train):
        return parser

    def print_network(self):
        n = sum(p.numel() for p in self.parameters())
        print("Network [%s] was created. Total number of parameters: %.1f million."
              % (type(self).__name__, n / 1e6))

    def init_weights(self, init_type="xavier", gain=0.02):
        """SPADE convention: xavier-normal(gain) on conv / linear weights, zero biases."""
        with torch.no_grad():
            for name, p in self.named_parameters():
                if name.endswith(".bias"):
                    p.zero_()
                    continue
                if p.dim() < 2:
                    continue
                tmp = torch.empty(p.shape, dtype=torch.float32)
                if init_type == "normal":
                    nn.init.normal_(tmp, 0.0, gain)
                elif init_type == "xavier":
                    nn.init.xavier_normal_(tmp, gain=gain)
                elif init_type == "xavier_uniform":
                    nn.init.xavier_uniform_(tmp, gain=1.0)
                elif init_type == "kaiming":
                    nn.init.kaiming_normal_(tmp, a=0, mode="fan_in")
                elif init_type == "orthogonal":
                    nn.init.orthogonal_(tmp, gain=gain)
                elif init_type == "none":
                    continue
                else:
                    raise NotImplementedError("initialization method [%s] is not implemented" % init_type)
                p.copy_(tmp)
        if self.finalized:
            self.store.repack()

    # ---- MI355X hooks --------------------------------------------------------------------------------
    def finalize(self, device, compute_dtype):
        """Move the parameters into the flat HBM store on `device` and build the packed compute-dtype operands.
        Must be called once, after construction / weight loading, before forward."""
        if device.type != "cuda":
            raise RuntimeError("the S2P hot path runs on a HIP device only (no CPU fallback); got device %s" % device)
        self.compute_dtype = compute_dtype
        self._declare_packs(compute_dtype)
        self.store.finalize(device)
        self.store.param_names = {id(p): n for n, p in self.named_parameters()}     # part of the checkpointed layout signature
        self.finalized = True
        return self

    def _declare_packs(self, compute_dtype):
        raise NotImplementedError

    def load_state_dict(self, state_dict, strict=True):
        out = super().load_state_dict(state_dict, strict=strict)
        if self.finalized:
            self.store.repack()
        return out

    def export_state_dict(self):
        """state_dict with plain contiguous CPU tensors (torch-layout), independent of the flat store."""
        return {k: v.detach().cpu().contiguous().clone() for k, v in self.state_dict().items()}

    def _require_ready(self):
        if not self.finalized:
            raise RuntimeError(f"{type(self).__name__}: call .finalize(device, dtype) before forward "
                               "(the HIP path has no CPU fallback)")

"""Pix2PixModel-style wrapper (SPADE lineage, README.md:73): `forward(data, mode)` with
modes generator / discriminator / inference, `create_optimizers`, `save` / load of
`<checkpoints_dir>/<env_type>_<epoch>.pth` (README.md:19-26).

data dict: {'prev_image': fp32 [N,3,H,W] in [-1,1], 'state': fp32 [N,S], 'image': fp32 [N,3,H,W]}.
"""
import os

import torch
import torch.nn as nn

from . import networks
from . import autograd_nodes
from .. import ops as ops_mod
from .autograd_nodes import d_step_apply, g_losses_apply, generator_apply, nhwc_to_nchw_apply, vgg_real_prefetch
from .networks.loss import VGG19


class FlatAdam:
    """torch.optim-like facade over ParamStore's fused HIP Adam (one launch over the flat buffer)."""

    def __init__(self, net, lr, betas, eps=1e-8):
        self.net, self.lr, self.betas, self.eps = net, lr, betas, eps
        self.grad_scale = 1.0
        self.param_groups = [dict(lr=lr)]

    def zero_grad(self, set_to_none=False):
        self.net.store.zero_grad()

    def step(self):
        self.net.store.adam_step(self.param_groups[0]["lr"], self.betas[0], self.betas[1], self.eps, self.grad_scale)

    def step_early(self, start):
        """This step's update of the flat range [start, end), whose gradients are already final (called from inside the backward,
        on a side stream); `step()` then finishes [0, start) and repacks."""
        self.net.store.adam_step_early(start, self.param_groups[0]["lr"], self.betas[0], self.betas[1], self.eps, self.grad_scale)

    def state_dict(self):
        return dict(lr=self.param_groups[0]["lr"], **self.net.store.optimizer_state())

    def load_state_dict(self, st, allow_unsigned=False):
        self.param_groups[0]["lr"] = st.get("lr", self.lr)
        self.net.store.load_optimizer_state(st, allow_unsigned=allow_unsigned)


class Pix2PixModel(nn.Module):
    @staticmethod
    def modify_commandline_options(parser, is_train):
        return parser

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        if not opt.gpu_ids:
            raise RuntimeError("S2P runs on a HIP device only: --gpu_ids -1 (CPU) is not supported; there is no CPU "
                               "fallback in the product path (tests/ use oracle/ for CPU references)")
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible (torch.cuda.is_available() is False): the S2P hot path cannot run")
        self.device = torch.device("cuda", opt.gpu_ids[0])
        torch.cuda.set_device(self.device)
        self.compute_dtype = torch.bfloat16 if opt.precision == "bf16" else torch.float32
        # False (default): the loss nodes scale their gradient seeds by whatever upstream gradient autograd hands them
        # (loss weights, .mean() over several terms, ...).  Pix2PixTrainer sets it to True because its
        # `sum(losses.values()).mean().backward()` passes exactly 1 to every term, which saves ~12 tiny launches a step.
        self.assume_unit_loss_grad = False
        self.before_netD = None                 # optional callable run just before netD's forward (data-parallel overlap)
        self.netG, self.netD, self.vgg = self.initialize_networks(opt)

    # ---- construction / checkpoint I/O ---------------------------------------------------------------------
    def ckpt_path(self, epoch):
        return os.path.join(self.opt.checkpoints_dir, "%s_%s.pth" % (self.opt.env_type, epoch))

    def initialize_networks(self, opt):
        netG = networks.define_G(opt)
        netD = networks.define_D(opt) if opt.isTrain else None
        vgg = None
        ckpt = None
        want = (not opt.isTrain) or getattr(opt, "continue_train", False)
        if want:
            path = self.ckpt_path(opt.which_epoch)
            if os.path.exists(path):
                ckpt = torch.load(path, map_location="cpu")
            elif not getattr(opt, "random_init", False):
                # also for --continue_train: silently restarting from random weights would overwrite the run's files
                raise FileNotFoundError("%s not found (README: put <env_type>_<epoch>.pth under --checkpoints_dir), "
                                        "or pass --random_init" % path)
        if ckpt is not None:
            netG.load_state_dict(ckpt["netG"] if "netG" in ckpt else ckpt)
            if netD is not None and "netD" in ckpt:
                netD.load_state_dict(ckpt["netD"])
        netG.finalize(self.device, self.compute_dtype)
        if netD is not None:
            netD.finalize(self.device, self.compute_dtype)
            if not opt.no_vgg_loss:
                vgg = VGG19()
                if opt.vgg_weights:
                    vgg.load_torchvision(opt.vgg_weights)
                else:
                    vgg.init_standin()
                vgg.finalize(self.device, self.compute_dtype)
        self._resume = ckpt
        return netG, netD, vgg

    def save(self, epoch):
        os.makedirs(self.opt.checkpoints_dir, exist_ok=True)
        ck = dict(netG=self.netG.export_state_dict(), epoch=epoch, env_type=self.opt.env_type,
                  state_dim=self.opt.state_dim, epochs_done=getattr(self, "epochs_done", 0),
                  iters_done=getattr(self, "iters_done", 0))
        if self.netD is not None:
            ck["netD"] = self.netD.export_state_dict()
        opts = getattr(self, "_optimizers", None)
        if opts:
            ck["optG"], ck["optD"] = opts[0].state_dict(), opts[1].state_dict()
        torch.save(ck, self.ckpt_path(epoch))

    def create_optimizers(self, opt):
        if opt.no_TTUR:
            beta1, beta2 = opt.beta1, opt.beta2
            G_lr, D_lr = opt.lr, opt.lr
        else:
            beta1, beta2 = 0.0, 0.9
            G_lr, D_lr = opt.lr / 2, opt.lr * 2
        optG = FlatAdam(self.netG, G_lr, (beta1, beta2))
        optD = FlatAdam(self.netD, D_lr, (beta1, beta2))
        if self._resume is not None and "optG" in self._resume:
            unsigned_ok = bool(getattr(opt, "allow_unsigned_optimizer_state", False))
            optG.load_state_dict(self._resume["optG"], allow_unsigned=unsigned_ok)
            optD.load_state_dict(self._resume["optD"], allow_unsigned=unsigned_ok)
        self._optimizers = (optG, optD)
        return optG, optD

    # ---- forward ---------------------------------------------------------------------------------------------
    def preprocess_input(self, data):
        """Device fp32 copies of the batch.  The G step and the D step of one iteration are handed the same batch: the
        copies are made once (keyed on the source tensors' identity and in-place version), which also lets the D step
        recognise the inputs of the G step's netD(prev, real) pass (autograd_nodes.dreal_pass)."""
        dev = self.device
        src = (data["prev_image"], data["state"], data.get("image"))
        key = tuple((id(t), t._version, t.data_ptr()) if t is not None else None for t in src)
        hit = getattr(self, "_pp_cache", None)
        if hit is not None and hit[0] == key and all(a is b for a, b in zip(hit[1], src)):
            return hit[2]
        prev = src[0].to(dev, dtype=torch.float32, non_blocking=True).contiguous()
        state = src[1].to(dev, dtype=torch.float32, non_blocking=True).contiguous()
        real = src[2]
        if real is not None:
            real = real.to(dev, dtype=torch.float32, non_blocking=True).contiguous()
        self._pp_cache = (key, src, (prev, state, real))
        return prev, state, real

    def forward(self, data, mode):
        prev, state, real = self.preprocess_input(data)
        if mode == "generator":
            g_loss, generated = self.compute_generator_loss(prev, state, real)
            return g_loss, generated
        elif mode == "discriminator":
            return self.compute_discriminator_loss(prev, state, real)
        elif mode == "inference":
            with torch.no_grad():
                return self.netG(prev, state)
        raise ValueError("|mode| is invalid")

    def compute_generator_loss(self, prev, state, real):
        pre = None
        if not self.opt.no_vgg_loss and autograd_nodes.OVERLAP_VGG and not ops_mod.SERIALIZE:
            # VGG features of the real image: on the VGG side stream, under the generator forward below
            from .._lib import chunk_elems
            pre = vgg_real_prefetch(self, real, self.compute_dtype, chunk_elems(self.compute_dtype))
        pre_d = None
        if autograd_nodes.OVERLAP_DREAL and not ops_mod.SERIALIZE and self.before_netD is None:
            # D(prev, real): on its side stream, under the generator forward (with data parallelism netD's weight update of
            # the previous step may still be in flight here: the pass then runs inside the loss node, after the wait)
            pre_d = autograd_nodes.dreal_pass(self, prev, real, True)
        fake = generator_apply(self.netG, prev, state)                 # NHWC compute dtype, autograd node
        L = g_losses_apply(self, fake, prev, real, pre, pre_d)
        G_losses = {"GAN": L[0]}
        if not self.opt.no_ganFeat_loss:
            G_losses["GAN_Feat"] = L[1]
        if not self.opt.no_vgg_loss:
            G_losses["VGG"] = L[2]
        if self.opt.lambda_l1 > 0:
            G_losses["L1"] = L[3]
        self._last_fake = fake.detach()
        return G_losses, fake

    def compute_discriminator_loss(self, prev, state, real, reuse_fake=False):
        fake = self._last_fake if (reuse_fake and getattr(self, "_last_fake", None) is not None) else None
        L = d_step_apply(self, prev, state, real, fake)     # generator forward (no grad) + both D halves, one node
        return {"D_Fake": L[0], "D_real": L[1]}

    def generated_to_nchw(self, fake_nhwc):
        with torch.no_grad():
            return nhwc_to_nchw_apply(fake_nhwc.detach(), 3)

"""Pix2PixTrainer (SPADE lineage): `run_generator_one_step` / `run_discriminator_one_step`, LR schedule, save.

MI355X-first data parallelism: one process per GPU; after each backward the network's whole flat gradient buffer is
averaged with ONE RCCL all-reduce (s2p_amd.parallel), then one fused Adam launch.  No DataParallel wrapper.
"""
import torch

from ..models.pix2pix_model import Pix2PixModel
from .. import parallel


class Pix2PixTrainer:
    def __init__(self, opt):
        self.opt = opt
        self.pix2pix_model = Pix2PixModel(opt)
        self.pix2pix_model_on_one_gpu = self.pix2pix_model
        # both steps below call sum(losses).mean().backward(): every loss term's upstream gradient is exactly 1
        self.pix2pix_model.assume_unit_loss_grad = True
        self.generated = None
        self.dp = parallel.DataParallelGroup.from_env()
        if opt.isTrain:
            self.dp.broadcast_store(self.pix2pix_model.netG.store)
            self.dp.broadcast_store(self.pix2pix_model.netD.store)
            self.optimizer_G, self.optimizer_D = self.pix2pix_model.create_optimizers(opt)
            self.optimizer_G.grad_scale = self.optimizer_D.grad_scale = 1.0 / self.dp.world_size
            # resume (--continue_train): the schedule continues from the restored optimizer lr and epoch counter
            ck = self.pix2pix_model._resume
            self.first_epoch = int(ck.get("epochs_done", 0)) + 1 if ck is not None else 1
            self.pix2pix_model.epochs_done = self.first_epoch - 1
            self.pix2pix_model.iters_done = int(ck.get("iters_done", 0)) if ck is not None else 0
            g_lr = self.optimizer_G.param_groups[0]["lr"]
            self.old_lr = g_lr if opt.no_TTUR else g_lr * 2
        self.g_losses, self.d_losses = {}, {}

    def run_generator_one_step(self, data):
        self.optimizer_G.zero_grad()
        g_losses, generated = self.pix2pix_model(data, mode="generator")
        g_loss = sum(g_losses.values()).mean()
        g_loss.backward()
        self.dp.all_reduce_grads(self.pix2pix_model.netG.store)
        self.optimizer_G.step()
        self.g_losses = g_losses
        self.generated = generated

    def run_discriminator_one_step(self, data):
        self.optimizer_D.zero_grad()
        d_losses = self.pix2pix_model(data, mode="discriminator")
        d_loss = sum(d_losses.values()).mean()
        d_loss.backward()
        self.dp.all_reduce_grads(self.pix2pix_model.netD.store)
        self.optimizer_D.step()
        self.d_losses = d_losses

    def get_latest_losses(self):
        return {**self.g_losses, **self.d_losses}

    def get_latest_generated(self):
        return self.pix2pix_model.generated_to_nchw(self.generated)

    def save(self, epoch):
        if self.dp.rank == 0:
            self.pix2pix_model.save(epoch)

    def end_of_epoch(self, epoch, iters):
        """Record progress so that a checkpoint written now resumes at epoch + 1."""
        self.pix2pix_model.epochs_done = epoch
        self.pix2pix_model.iters_done = iters

    def update_learning_rate(self, epoch):
        opt = self.opt
        if epoch > opt.niter:
            lrd = opt.lr / max(opt.niter_decay, 1)
            new_lr = max(self.old_lr - lrd, 0.0)
        else:
            new_lr = self.old_lr
        if new_lr != self.old_lr:
            if opt.no_TTUR:
                g, d = new_lr, new_lr
            else:
                g, d = new_lr / 2, new_lr * 2
            self.optimizer_G.param_groups[0]["lr"] = g
            self.optimizer_D.param_groups[0]["lr"] = d
            print("update learning rate: %f -> %f" % (self.old_lr, new_lr))
            self.old_lr = new_lr

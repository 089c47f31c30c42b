"""Pix2PixTrainer (SPADE lineage): `run_generator_one_step` / `run_discriminator_one_step`, LR schedule, save.

MI355X-first data parallelism: one process per GPU; the network's whole gradient lives in ONE flat buffer, so the
exchange is a couple of large RCCL all-reduces per step (xGMI is point-to-point: few large messages use every link),
followed by one fused Adam launch.  No DataParallel wrapper, no per-tensor buckets.  With more than one rank the
collectives run on a communication stream and are hidden under compute:

  * G step: the flat G buffer is laid out so that the 97 % of the gradient that is final once the ResBlk chain and the
    deferred weight gradients are done is one contiguous tail; it is exchanged in THREE buckets that follow the batches of
    deferred weight gradients (two ResBlks + the gamma/beta heads of their four norms each: 20 / 19 / 19 MB): a bucket's
    all-reduce starts from a hook inside the generator's backward as soon as its batch has finished, so two thirds of
    the payload are in flight before the last weight-gradient launch and the last third runs under the rest of the backward
    (gamma/beta dgrad, shared conv, encoder, state path); only the small head of the buffer is reduced after the backward;
  * D step: all-reduce + Adam + weight repack of D run entirely on the communication stream, under the NEXT step's
    generator forward; the next G step waits for them right before its first use of D (a hook in the G-loss node).

`seg` (optional `StepGraph`): when the step is captured into hipGraphs, every collective / event / cross-stream wait
is a `cut` point between graph segments; in eager mode a cut just runs its action.
"""
import torch
import torch.distributed as dist

from ..models.pix2pix_model import Pix2PixModel
from .. import parallel


EARLY_ADAM = False      # one rank: the generator's Adam update of the early-complete tail under the rest of the backward instead of after it.
                        # Off: same-box A/B 8.957 (on) vs 8.939 ms/step (off) -- the update only trades HBM time with the backward's own passes


class Pix2PixTrainer:
    def __init__(self, opt):
        self.opt = opt
        self.pix2pix_model = Pix2PixModel(opt)
        self.pix2pix_model_on_one_gpu = self.pix2pix_model
        # both steps below call sum(losses).mean().backward(): every loss term's upstream gradient is exactly 1
        self.pix2pix_model.assume_unit_loss_grad = True
        self.generated = None
        self.seg = None
        self._eD = None                 # event: D's all-reduce + Adam + repack of the previous step are done
        self._eD_waited = True          # ... and whether the main stream has waited for it since it was recorded
        self._buckets = None            # S2PGenerator.early_buckets(), resolved at the first exchange
        self.comm_events = None         # a list: (event, event) pairs around every wait of the compute stream for the exchange
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
            if self.dp.active:
                self.pix2pix_model.netG.on_early_grads = self._allreduce_G_tail
        self.g_losses, self.d_losses = {}, {}

    # ---- cut points (collectives, events, cross-stream waits stay outside captured graphs) --------------------------------
    def _cut(self, action):
        if self.seg is not None:
            self.seg.cut(action)
        else:
            action()

    def _timed_wait(self, action):
        """`action` makes the current stream wait for the communication stream; with `comm_events` set (bench.py's instrumented
        step) the wait is bracketed by two events on the current stream: their distance is the EXPOSED communication time."""
        if self.comm_events is None:
            return action()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        action()
        e1.record()
        self.comm_events.append((e0, e1))

    def _wait_D_update(self):
        def act():
            if self._eD is not None:
                self._timed_wait(lambda: torch.cuda.current_stream().wait_event(self._eD))
        self._cut(act)
        self._eD_waited = True

    def _allreduce_G_tail(self, bucket=None, after=None):
        """Hook of the generator's backward: `bucket` (see S2PGenerator.early_buckets) of the flat gradient's early-complete tail is
        final -- start its all-reduce on the communication stream, under the rest of the backward.  bucket None: the whole tail."""
        net = self.pix2pix_model.netG
        g = net.store.grad
        if bucket is None:
            parts = [g[net.early_grad_offset:]]
        else:
            if self._buckets is None:
                self._buckets = net.early_buckets()
            parts = [g[o:o + n] for o, n in self._buckets[bucket]]

        def act():
            for t in parts:
                self.dp.all_reduce_async(t, after=after)     # `after`: an event of the weight-gradient side stream (eager mode)
        self._cut(act)

    def _finish_G_exchange(self):
        net = self.pix2pix_model.netG
        head = net.store.grad[:net.early_grad_offset]

        def act():
            self.dp.all_reduce_async(head)
            self._timed_wait(self.dp.join)
        self._cut(act)

    def _finish_D_async(self):
        """All-reduce, Adam and repack of D on the communication stream; `_eD` marks their completion."""
        comm = self.dp.comm_stream()
        comm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(comm):
            dist.all_reduce(self.pix2pix_model.netD.store.grad, op=dist.ReduceOp.SUM, group=self.dp.group)
            if not (self.seg is not None and self.seg.capturing):
                # (while the step is being captured its kernels are only recorded: the gradients in the buffer are the
                # previous step's, already applied -- a second update with them would be a spurious Adam step)
                self.optimizer_D.step()
            ev = torch.cuda.Event()
            ev.record(comm)
        self._eD = ev
        self._eD_waited = False

    # ---- the two steps ---------------------------------------------------------------------------------------------------
    def run_generator_one_step(self, data):
        multi = self.dp.active
        self.optimizer_G.zero_grad()
        self.pix2pix_model.before_netD = self._wait_D_update if multi else None
        g_losses, generated = self.pix2pix_model(data, mode="generator")
        self.pix2pix_model.before_netD = None
        netG = self.pix2pix_model.netG
        if not multi and EARLY_ADAM:
            # one rank: Adam on the early-complete tail of the flat buffer from inside the backward (S2PGenerator.on_tail_final)
            netG.on_tail_final = lambda: self.optimizer_G.step_early(netG.early_grad_offset)
        try:
            self._backward(g_losses)       # multi-rank: the bucket all-reduces are launched from inside (on_early_grads)
        finally:
            netG.on_tail_final = None
        if multi:
            self._finish_G_exchange()
        self.optimizer_G.step()
        self.g_losses = g_losses
        self.generated = generated

    def run_discriminator_one_step(self, data):
        if self.dp.active and not self._eD_waited:
            # the previous D step's all-reduce + Adam + repack may still be running on the communication stream: they read
            # netD.store.grad and rewrite the packed weights, which zero_grad / the forward below touch.  Two D steps in a row
            # (--D_steps_per_G > 1) get here un-waited; after a G step the hook in the G-loss node has already waited.
            self._wait_D_update()
        self.optimizer_D.zero_grad()
        d_losses = self.pix2pix_model(data, mode="discriminator")
        self._backward(d_losses)
        if self.dp.active:
            self._cut(self._finish_D_async)
        else:
            self.optimizer_D.step()
        self.d_losses = d_losses

    def _backward(self, losses):
        """sum(losses.values()).mean().backward() without the adds / mean / their autograd kernels (~10 tiny launches per
        call): every loss term is a 0-dim tensor and receives the upstream gradient 1 directly."""
        vals = list(losses.values())
        one = getattr(self, "_one", None)
        if one is None or one.device != vals[0].device:
            one = self._one = torch.ones((), dtype=vals[0].dtype, device=vals[0].device)
        torch.autograd.backward(vals, [one] * len(vals))

    def sync(self):
        """Wait (on the current stream) for everything the trainer has in flight on the communication stream."""
        self.dp.join()

    def get_latest_losses(self):
        return {**self.g_losses, **self.d_losses}

    def get_latest_generated(self):
        return self.pix2pix_model.generated_to_nchw(self.generated)

    def save(self, epoch):
        self.sync()
        if self.dp.rank == 0:
            self.pix2pix_model.save(epoch)

    def end_of_epoch(self, epoch, iters):
        """Record progress so that a checkpoint written now resumes at epoch + 1."""
        self.pix2pix_model.epochs_done = epoch
        self.pix2pix_model.iters_done = iters

    def update_learning_rate(self, epoch):
        opt = self.opt
        self.sync()                         # the D optimizer may still be stepping on the communication stream
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

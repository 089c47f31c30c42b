"""train.py -- S2P training (README.md:59):
    python train.py --dataroot=./datasets/cheetah.hdf5 --env_type=cheetah --netG=s2p --batchSize=16 --gpu_ids=0
Multi-GPU: one process per GPU,  python -m torch.distributed.run --nproc-per-node 8 train.py ...  (RCCL all-reduce of
the flat G/D gradient buffers)."""
import time

import torch

from s2p_amd.data import create_dataloader
from s2p_amd.options.train_options import TrainOptions
from s2p_amd.trainers.pix2pix_trainer import Pix2PixTrainer


def main(args=None):
    opt = TrainOptions().parse(args, save=True)
    trainer = Pix2PixTrainer(opt)
    dl = create_dataloader(opt, trainer.dp.rank, trainer.dp.world_size)
    total_epochs = opt.niter + opt.niter_decay
    it = trainer.pix2pix_model.iters_done                 # > 0 when resuming with --continue_train
    # --hip_graph: the G+D step is captured once into hipGraph segments (s2p_amd/stepgraph.py) and replayed; each batch is
    # copied into static device buffers first.  The capture bakes the learning rate in, so it is redone when the rate changes.
    sg, static, captured_lr, warned_graph = None, None, None, False
    for epoch in range(trainer.first_epoch, total_epochs + 1):
        if hasattr(dl.sampler, "set_epoch"):
            dl.sampler.set_epoch(epoch)
        t0 = time.time()
        for i, data in enumerate(dl):
            it += 1
            if opt.hip_graph and opt.D_steps_per_G == 1:
                from s2p_amd.stepgraph import StepGraph
                dev = trainer.pix2pix_model.device
                if static is None:
                    static = {k: data[k].to(dev, torch.float32).contiguous().clone() for k in ("prev_image", "state", "image")}
                for k in static:
                    static[k].copy_(data[k], non_blocking=True)
                trained = False
                if sg is None or captured_lr != trainer.old_lr:
                    def train_step():
                        trainer.run_generator_one_step(static)
                        trainer.run_discriminator_one_step(static)
                    if sg is None:
                        train_step()                       # the very first batch is trained on eagerly (allocator / library warm-up)
                        trained = True
                    sg = StepGraph()
                    trainer.seg = sg
                    sg.capture(train_step)                 # kernels are only RECORDED during capture: no update happens here
                    captured_lr = trainer.old_lr
                if not trained:
                    sg.replay()                            # every other batch (also the one a re-capture saw) is trained on by a replay
            else:
                if opt.hip_graph and not warned_graph and trainer.dp.rank == 0:
                    print("warning: --hip_graph is ignored with --D_steps_per_G %d (the captured step is one G + one D step)" % opt.D_steps_per_G)
                    warned_graph = True
                if i % opt.D_steps_per_G == 0:
                    trainer.run_generator_one_step(data)
                trainer.run_discriminator_one_step(data)
            if it % opt.print_freq == 0 and trainer.dp.rank == 0:
                losses = {k: float(v) for k, v in trainer.get_latest_losses().items()}
                print("(epoch %d, iters %d) " % (epoch, it) + " ".join("%s: %.3f" % kv for kv in losses.items()))
        trainer.update_learning_rate(epoch)
        trainer.end_of_epoch(epoch, it)
        if trainer.dp.rank == 0:
            print("End of epoch %d / %d \t Time Taken: %d sec" % (epoch, total_epochs, time.time() - t0))
        if epoch % opt.save_epoch_freq == 0 or epoch == total_epochs:
            trainer.save("latest")
            trainer.save(epoch)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()

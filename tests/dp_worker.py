"""Child process of tests/test_dp_gpu.py: one data-parallel rank running ONE real Pix2PixTrainer G step + D step.
Usage: python dp_worker.py RANK WORLD PORT OUTFILE [rccl | d2]   (DP_B: global batch, default 4; all ranks share GPU 0; gloo carries the collectives, so the
N>1 code path of trainer / parallel.py / FlatAdam runs on a one-GPU box exactly as it does over RCCL.  With `rccl` and
WORLD 1 the exchange path runs over a one-rank RCCL group instead: S2P_FORCE_DP, backend 'nccl')."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    rccl = len(sys.argv) > 5 and sys.argv[5] == "rccl"
    two_d = len(sys.argv) > 5 and sys.argv[5] == "d2"       # --D_steps_per_G 2: a second D step right behind the first
    if rccl:
        assert world == 1
        os.environ["S2P_FORCE_DP"] = "1"                    # DataParallelGroup.from_env makes the one-rank 'nccl' group
    import torch
    import torch.distributed as dist
    from s2p_amd.options.train_options import TrainOptions
    from s2p_amd.trainers.pix2pix_trainer import Pix2PixTrainer
    if world > 1:
        dist.init_process_group(backend="gloo")
    B = int(os.environ.get("DP_B", "4"))                    # global batch; each rank takes B / world samples
    per = B // world
    opt = TrainOptions().parse(["--env_type", "cheetah", "--batchSize", str(per), "--precision", "bf16", "--gpu_ids", "0",
                                "--checkpoints_dir", os.path.dirname(out)], quiet=True)
    torch.manual_seed(7 + rank)                             # ranks start from DIFFERENT weights: the broadcast must fix that
    tr = Pix2PixTrainer(opt)
    model = tr.pix2pix_model
    g = torch.Generator().manual_seed(99)
    prev = torch.rand(B, 3, 84, 84, generator=g) * 2 - 1
    real = torch.rand(B, 3, 84, 84, generator=g) * 2 - 1
    state = torch.randn(B, 17, generator=g)
    sl = slice(rank * per, (rank + 1) * per)
    data = dict(prev_image=prev[sl], state=state[sl], image=real[sl])
    w0 = model.netG.store.master.detach().cpu().clone()
    tr.run_generator_one_step(data)
    gG = (model.netG.store.grad * tr.optimizer_G.grad_scale).detach().cpu().clone()
    wG = model.netG.store.master.detach().cpu().clone()
    # The D step makes its fake with the generator's weights.  Adam's first step is sign-like (lr * g / (|g| + eps)), so the
    # just-updated G weights of the two runs differ by up to lr wherever a gradient is ~0 (e.g. the conv biases in front of
    # an InstanceNorm), which moves bf16 roundings of the fake and flips LeakyReLU branches in D.  Put the initial G
    # weights back so that the D-step comparison sees identical inputs in both runs.
    model.netG.store.master.copy_(w0.to(model.netG.store.master.device))
    model.netG.store.repack()
    tr.run_discriminator_one_step(data)
    if two_d:
        # the first D step's all-reduce + Adam + repack are still running on the communication stream: the second step's
        # zero_grad / forward must wait for them
        tr.run_discriminator_one_step(data)
    tr.sync()                                               # D's all-reduce + Adam run on the communication stream
    gD = (model.netD.store.grad * tr.optimizer_D.grad_scale).detach().cpu().clone()
    torch.cuda.synchronize()
    losses = {k: float(v) for k, v in tr.get_latest_losses().items()}
    torch.save(dict(w0=w0, gG=gG, gD=gD, wG=wG, wD=model.netD.store.master.detach().cpu(),
                    losses=losses, world=tr.dp.world_size, active=bool(tr.dp.active),
                    backend=dist.get_backend() if dist.is_initialized() else None, lrG=tr.optimizer_G.param_groups[0]["lr"],
                    lrD=tr.optimizer_D.param_groups[0]["lr"]), out)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

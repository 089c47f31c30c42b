"""CPU, world_size 2, gloo: the N>1 path of the train step -- flat-gradient all-reduce, parameter broadcast and
max-over-ranks timing (s2p_amd/parallel.py).  On GPUs the same code runs over RCCL (backend 'nccl')."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class _FakeStore:
    def __init__(self, rank):
        g = torch.Generator().manual_seed(100 + rank)
        self.master = torch.randn(1003, generator=g)
        self.grad = torch.randn(1003, generator=g)
        self.repacked = 0

    def repack(self):
        self.repacked += 1


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from s2p_amd.parallel import DataParallelGroup
    dp = DataParallelGroup.from_env(backend="gloo")
    st = _FakeStore(rank)
    g_local = st.grad.clone()
    dp.broadcast_store(st)
    dp.all_reduce_grads(st)
    t = dp.max_over_ranks(1.0 + rank)
    dp.barrier()
    # numpy copies: a torch tensor in a multiprocessing queue is rebuilt through the SENDER's socket, which is gone once it exits
    q.put((rank, st.master.numpy().copy(), st.grad.numpy().copy(), g_local.numpy().copy(), t, st.repacked, dp.world_size))
    dist.destroy_process_group()


def test_flat_grad_allreduce_and_broadcast_world8():
    """The world size the driver's scaling run ends at: SUM over 8 contributors, broadcast from rank 0 to 7 others, max over 8,
    and the 1/8 that Adam's grad_scale applies to the sum."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    W = 8
    procs = [ctx.Process(target=_worker, args=(r, W, port, q)) for r in range(W)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(W)], key=lambda r: r[0])
    res = [(r[0], torch.from_numpy(r[1]), torch.from_numpy(r[2]), torch.from_numpy(r[3]), r[4], r[5], r[6]) for r in res]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    total = sum(r[3] for r in res)
    for (rank, m, g, l, t, rp, ws) in res:
        assert ws == W and rp == 1 and t == float(W)            # max over ranks of (1 + rank)
        assert torch.equal(m, _FakeStore(0).master)             # everyone holds rank 0's parameters
        assert torch.equal(g, res[0][2])                        # identical reduced buffer on every rank
        assert torch.allclose(g, total, rtol=1e-6, atol=1e-6)
    mean = res[0][2] * (1.0 / W)                                # what the fused Adam sees (grad_scale = 1 / world)
    assert torch.allclose(mean, total / W, rtol=1e-6, atol=1e-6)


def test_flat_grad_allreduce_and_broadcast_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda r: r[0])
    res = [(r[0], torch.from_numpy(r[1]), torch.from_numpy(r[2]), torch.from_numpy(r[3]), r[4], r[5], r[6]) for r in res]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, m0, g0, l0, t0, rp0, ws0), (r1, m1, g1, l1, t1, rp1, ws1) = res
    assert ws0 == ws1 == 2
    assert torch.equal(m0, m1)                                  # broadcast from rank 0
    assert torch.equal(m0, _FakeStore(0).master)
    assert torch.allclose(g0, l0 + l1) and torch.equal(g0, g1)  # SUM all-reduce (1/world folded into Adam's grad_scale)
    assert t0 == t1 == 2.0                                      # max over ranks
    assert rp0 == rp1 == 1


def test_single_process_group_is_a_noop():
    from s2p_amd.parallel import DataParallelGroup
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        os.environ.pop(k, None)
    dp = DataParallelGroup.from_env()
    assert dp.world_size == 1 and dp.rank == 0
    st = _FakeStore(0)
    g = st.grad.clone()
    dp.all_reduce_grads(st); dp.broadcast_store(st); dp.barrier()
    assert torch.equal(st.grad, g) and dp.max_over_ranks(3.5) == 3.5

"""Data parallelism for the S2P train step: one process per GPU, RCCL over xGMI (torch.distributed 'nccl' backend is
RCCL on ROCm).  The path shards over the batch with a single real exchange step per network per step: the average of
the flat gradient buffer (G: ~67 MB, D: ~22 MB fp32).  Because all gradients of a network already live in ONE
contiguous buffer (ParamStore.grad), the exchange is ONE all-reduce call per network -- no bucketing logic, and on
the 7-link xGMI mesh a single large message is what lets RCCL use every link.
"""
import os

import torch
import torch.distributed as dist


class DataParallelGroup:
    def __init__(self, rank=0, world_size=1, group=None):
        self.rank, self.world_size, self.group = rank, world_size, group

    @classmethod
    def from_env(cls, backend=None):
        ws = int(os.environ.get("WORLD_SIZE", "1"))
        if ws <= 1:
            return cls()
        if not dist.is_initialized():
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend=backend)
        return cls(dist.get_rank(), dist.get_world_size())

    def all_reduce_(self, flat):
        """Sum `flat` over ranks in place (the 1/world factor is folded into the fused Adam's grad_scale)."""
        if self.world_size > 1:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        return flat

    def all_reduce_grads(self, store):
        return self.all_reduce_(store.grad)

    # ---- overlapped exchange: collectives on a communication stream, ordered against compute with events ---------
    def comm_stream(self):
        s = getattr(self, "_comm", None)
        if s is None:
            s = self._comm = torch.cuda.Stream()
        return s

    def all_reduce_async(self, flat):
        """Sum `flat` over ranks on the communication stream, after everything issued so far on the current stream.
        Compute issued later on the current stream runs concurrently; call `join()` before it reads `flat`."""
        if self.world_size <= 1:
            return
        comm = self.comm_stream()
        comm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(comm):
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)

    def join(self):
        """Make the current stream wait for everything issued on the communication stream."""
        if self.world_size > 1:
            torch.cuda.current_stream().wait_stream(self.comm_stream())

    def broadcast_store(self, store, src=0):
        """Make every rank start from rank `src`'s parameters (C2 in SURVEY.md section 2.2)."""
        if self.world_size > 1:
            dist.broadcast(store.master, src=src, group=self.group)
            store.repack()

    def barrier(self):
        if self.world_size > 1:
            dist.barrier(group=self.group)

    def max_over_ranks(self, value):
        if self.world_size <= 1:
            return value
        dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
        t = torch.tensor([value], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

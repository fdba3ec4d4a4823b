"""torch.distributed (gloo on CPU) stand-in for mixmogam_amd.dist.RcclCollectives, same interface.
TEST INFRASTRUCTURE: the product package never imports torch."""
import numpy as np


class TorchCollectives(object):
    """torch.distributed (gloo on CPU) stand-in with the same interface, for tests."""

    def __init__(self):
        import torch.distributed as dist
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def allreduce(self, arr, op="sum"):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(arr).copy())
        self.dist.all_reduce(t, op={"sum": self.dist.ReduceOp.SUM, "min": self.dist.ReduceOp.MIN,
                                    "max": self.dist.ReduceOp.MAX}[op])
        return t.numpy()

    def allgather(self, arr):
        import torch
        t = torch.from_numpy(np.ascontiguousarray(arr))
        outs = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t)
        return np.concatenate([o.numpy() for o in outs])

    allgather_host = allgather

    @property
    def device_comm(self):
        return self          # tests/fake_ctx.py's stand-ins reduce through this object itself

    def barrier(self):
        self.dist.barrier()

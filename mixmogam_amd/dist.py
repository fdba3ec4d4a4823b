"""Multi-GPU sharding of the hot path: one process per GPU, SNP blocks partitioned across ranks.

SURVEY 8e: kinship shards the contraction (SNP) axis -> all-reduce SUM of the N x N integer
count matrix; eigh + REML are replicas; the EMMAX scan shards independent SNP blocks -> all-gather
of (rss, F, p); the permutation test shards SNP blocks -> all-reduce MIN over the P minima.

The collectives are behind a small interface (rank, world, allreduce, barrier) so that the same sharding
logic runs over RCCL on GPUs (RcclCollectives: libmixmogam_hip's mmg_comm_* over xGMI) and over gloo on CPU
in the world_size-2 tests (tests/torch_coll.py -- test infrastructure; nothing here imports torch).
"""
import numpy as np


def shard_range(total, rank, world):
    """Contiguous block [m0, m1) of `total` units owned by `rank` (sizes differ by at most 1)."""
    base, rem = divmod(int(total), int(world))
    m0 = rank * base + min(rank, rem)
    return m0, m0 + base + (1 if rank < rem else 0)


def file_bootstrap(rank, world, timeout_s=120.0):
    """Single-node bootstrap of rank 0's 128-byte ncclUniqueId through a file in /tmp, keyed by
    the launcher's MASTER_PORT / run id (torchrun exports them).  Returns a `bcast(raw)`
    callable for RcclCollectives.  No torch in the GPU process: the HIP library links the
    system ROCm runtime, and importing torch's bundled runtime beside it is not safe."""
    import os
    import time
    key = "%s_%s_%s_%d" % (os.environ.get("MASTER_ADDR", "local"), os.environ.get("MASTER_PORT", "0"),
                           os.environ.get("TORCHELASTIC_RUN_ID", "none"), world)
    if "TORCHELASTIC_RUN_ID" in os.environ:
        # the ranks of one torch.distributed.run launch are children of the same agent process: its pid
        # separates back-to-back launches that reuse a port (a stale file of a crashed run is never read)
        key += "_%d" % os.getppid()
    path = os.path.join("/tmp", "mmg_rdzv_" + "".join(c if c.isalnum() else "_" for c in key) + ".bin")
    t_start = time.time()

    def bcast(raw):
        if rank == 0:
            tmp = path + ".%d" % os.getpid()
            with open(tmp, "wb") as f:
                f.write(raw)
            os.replace(tmp, path)
            return raw
        while True:
            try:
                st = os.stat(path)
                if st.st_size == 128 and st.st_mtime > t_start - 30.0:
                    with open(path, "rb") as f:
                        return f.read()
            except OSError:
                pass
            if time.time() - t_start > timeout_s:
                raise RuntimeError("timed out waiting for rank 0's RCCL id at %s" % path)
            time.sleep(0.05)

    bcast.path = path
    return bcast


class RcclCollectives(object):
    """RCCL through the C ABI.  `bootstrap_bcast(bytes_or_None) -> bytes` distributes rank 0's
    ncclUniqueId (e.g. over torch.distributed gloo, a file, or MPI)."""

    def __init__(self, ctx, rank, world, bootstrap_bcast):
        import ctypes as C
        self.ctx, self.rank, self.world = ctx, rank, world
        uid = (C.c_ubyte * 128)()
        if rank == 0:
            ctx._check(ctx.lib.mmg_comm_unique_id(uid))
        raw = bootstrap_bcast(bytes(uid) if rank == 0 else None)
        uid = (C.c_ubyte * 128).from_buffer_copy(raw)
        h = C.c_void_p()
        ctx._check(ctx.lib.mmg_comm_create(ctx.h, uid, rank, world, C.byref(h)))
        self.h = h

    def allreduce(self, arr, op="sum"):
        from . import _lib
        code = {"sum": 0, "min": 1, "max": 2}[op]
        a = np.ascontiguousarray(arr)
        if a.dtype == np.int64:
            self.ctx._check(self.ctx.lib.mmg_comm_allreduce_i64(self.ctx.h, self.h, _lib._ptr(a), a.size, code))
        else:
            a = np.ascontiguousarray(a, dtype=np.float64)
            self.ctx._check(self.ctx.lib.mmg_comm_allreduce_f64(self.ctx.h, self.h, _lib._ptr(a), a.size, code))
        return a

    def allgather_scan(self, count):
        """all-gather the device-resident (rss, F, p) of the last scan; equal `count` per rank."""
        from . import _lib
        outs = [np.empty(self.world * count) for _ in range(3)]
        self.ctx._check(self.ctx.lib.mmg_comm_allgather_scan(self.ctx.h, self.h, count, *[_lib._ptr(o) for o in outs]))
        return outs

    def barrier(self):
        self.ctx._check(self.ctx.lib.mmg_comm_barrier(self.ctx.h, self.h))

    def close(self):
        if self.h is not None:
            self.ctx.lib.mmg_comm_destroy(self.ctx.h, self.h)
            self.h = None


def sharded_ibs_counts(local_counts, coll):
    """Partial IBS count matrices (one per rank, over that rank's SNP block) -> global counts.
    Integer all-reduce: exact and order independent."""
    return coll.allreduce(np.asarray(local_counts, dtype=np.int64), "sum")


def sharded_perm_min(local_min_rss, coll):
    return coll.allreduce(np.asarray(local_min_rss, dtype=np.float64), "min")


def pad_block(x, count):
    """Pad a per-rank result block to the common `count` (ragged last shard) with NaN."""
    out = np.full(count, np.nan)
    out[:len(x)] = x
    return out


def unpad_gathered(gathered, total, world):
    """Inverse of the rank-major gather of equal-sized padded blocks."""
    count = len(gathered) // world
    parts = []
    for r in range(world):
        m0, m1 = shard_range(total, r, world)
        parts.append(gathered[r * count:r * count + (m1 - m0)])
    return np.concatenate(parts)

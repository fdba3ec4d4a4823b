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


def file_bootstrap(rank, world, timeout_s=None):
    """Single-node bootstrap of rank 0's 128-byte ncclUniqueId through a file, keyed by the launcher's
    MASTER_PORT / run id (torchrun exports them).  Returns a `bcast(raw)` callable for RcclCollectives.
    No torch in the GPU process: the HIP library links the system ROCm runtime, and importing torch's bundled
    runtime beside it is not safe.

    The file lives in a per-user directory (mode 0700), is created with O_EXCL and mode 0600 after rank 0 has
    removed whatever a crashed earlier launch left under the same key, and is removed again by rank 0 once the
    communicator exists (RcclCollectives calls bcast.done()), so a relaunch on the same port never reads a stale
    id.  Ranks > 0 give up after MMG_RDZV_TIMEOUT seconds (default 120) -- e.g. when rank 0 died before writing --
    instead of entering ncclCommInitRank with nothing to meet."""
    import os
    import time
    if timeout_s is None:
        timeout_s = float(os.environ.get("MMG_RDZV_TIMEOUT", "120"))
    key = "%s_%s_%s_%s_%d" % (os.environ.get("MASTER_ADDR", "local"), os.environ.get("MASTER_PORT", "0"),
                              os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.environ.get("MMG_RUN_ID", "none"), world)
    if "TORCHELASTIC_RUN_ID" in os.environ:
        # the ranks of one torch.distributed.run launch are children of the same agent process: its pid
        # separates back-to-back launches that reuse a port
        key += "_%d" % os.getppid()
    rdir = os.path.join(os.environ.get("MMG_RDZV_DIR", "/tmp"), "mmg_rdzv_%d" % os.getuid())
    os.makedirs(rdir, mode=0o700, exist_ok=True)
    path = os.path.join(rdir, "".join(c if c.isalnum() else "_" for c in key) + ".bin")
    t_start = time.time()

    def bcast(raw):
        if rank == 0:
            try:
                os.unlink(path)
            except OSError:
                pass
            tmp = path + ".%d" % os.getpid()
            fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
            with os.fdopen(fd, "wb") as f:
                f.write(raw)
            os.replace(tmp, path)
            return raw
        while True:
            try:
                st = os.stat(path)
                if st.st_size == 128 and st.st_uid == os.getuid() and st.st_mtime > t_start - 30.0:
                    with open(path, "rb") as f:
                        return f.read()
            except OSError:
                pass
            if time.time() - t_start > timeout_s:
                raise RuntimeError("timed out after %.0f s waiting for rank 0's RCCL id at %s" % (timeout_s, path))
            time.sleep(0.05)

    def done():
        if rank == 0:
            try:
                os.unlink(path)
            except OSError:
                pass

    bcast.path = path
    bcast.done = done
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
        if hasattr(bootstrap_bcast, "done"):
            self.barrier()                      # every rank holds the id (it is inside its communicator) ...
            bootstrap_bcast.done()              # ... so rank 0 may remove the rendezvous file

    @property
    def device_comm(self):
        """What the sharded C-ABI entry points take as their communicator (reductions in HBM)."""
        return self.h

    def info(self):
        """(rank, world, ncclCommCount) as RCCL reports them."""
        import ctypes as C
        r, w, n = C.c_int(-1), C.c_int(-1), C.c_int(-1)
        self.ctx._check(self.ctx.lib.mmg_comm_info(self.h, C.byref(r), C.byref(w), C.byref(n)))
        return r.value, w.value, n.value

    def allgather(self, arr):
        """Equal-sized host blocks -> [world * len] (rank-major), staged through HBM and RCCL."""
        from . import _lib
        a = np.ascontiguousarray(arr, dtype=np.float64).reshape(-1)
        out = np.empty(self.world * a.size)
        self.ctx._check(self.ctx.lib.mmg_comm_allgather_f64(self.ctx.h, self.h, _lib._ptr(a), a.size, _lib._ptr(out)))
        return out

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

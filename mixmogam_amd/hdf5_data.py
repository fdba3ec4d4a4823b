"""Chunked EMMAX drivers -- /root/reference/hdf5_data.py calculate_ibd_kinship (:17-62), run_emmax (:70-187) and
run_emmax_perm (:191-351): MAF filter, GRM kinship accumulated over SNP chunks, REML once, EMMAX scan per
chromosome, optional permutation thresholds, results written in the reference's dataset layout.

Containers: the reference opens `h5py.File(hdf5_filename)`; here `chunkstore.open_container` opens the same tree
either from a directory of memory-mapped .npy datasets (no h5py needed, `raw_snps[i:j]` touches only those rows)
or, when h5py is installed and the path is a real HDF5 file, from h5py itself.  Every driver also accepts the
already-open tree / a plain mapping
    {chrom: {'raw_snps': int8 [M_c x N] (ndarray / memmap / h5py dataset), 'freqs': [M_c], 'positions': [M_c]}}
in place of the file name (then `phenotypes` is passed explicitly and nothing is written unless `out_file` is set).

Streaming: chunks are read from the source and uploaded by a helper thread on a second HIP context / stream of the
same device while the previous chunk is being computed on (PCIe ingest overlaps the kernels); nothing but the
current and the next chunk is resident, so a 500 GB genotype matrix streams through a fixed HBM footprint.

Multi-GPU (`coll`, mixmogam_amd.dist): chunks are dealt round-robin to the ranks; the kinship partial sums are
all-reduced in HBM (RCCL SUM), eigh/REML are replicated, every rank scans its chunks and the OWNED p-value blocks
are all-gathered; permutation minima are combined with a MIN/MAX all-reduce of P values.  Rank 0 writes the results.
"""
import os
import time

import numpy as np

from . import _lib, chunkstore, kinship
from . import linear_models as lm


def open_hdf5(filename, mode='r'):
    """The reference's genotype file as a chunk source: {'genot_data', 'phenotypes', 'indiv_ids', 'file'}."""
    f = chunkstore.open_container(filename, mode)
    ig = f['indiv_data']
    return {'genot_data': f['genot_data'], 'phenotypes': np.asarray(ig['phenotypes'][...]) if 'phenotypes' in ig else None,
            'indiv_ids': np.asarray(ig['indiv_ids'][...]), 'file': f}


_UPLOAD_CTX = {}
_POOLS = {}                # (device, N) -> (capacity in SNPs, [two chunk stores], [two page-locked staging buffers])


def release_pools():
    """Free the chunk stores and staging buffers the streaming loops keep between calls."""
    for _cap, stores, _host in _POOLS.values():
        for g in stores:
            g.close()
    _POOLS.clear()
    if _lib._default_ctx is not None:                                     # ... and what the default context keeps between calls
        _lib._default_ctx.trim()

# parallel readinto() streams per chunk (a single page-cache copy runs at 5-8 GB/s): a quarter of the host's cores, 4..16
_READ_THREADS = int(os.environ.get('MMG_READ_THREADS', 0)) or max(4, min(16, (os.cpu_count() or 8) // 4))


def _maf_filter(cg, min_maf):
    freqs = np.asarray(cg['freqs'][...], dtype=np.float64)
    return np.minimum(freqs, 1 - freqs) > min_maf                       # hdf5_data.py:91-93


def _raw_dataset(cg):
    """(dataset, bits): the chromosome's genotype rows -- `raw_snps` int8 [M_c x N] (plink2hdf5.py:111), or, where
    that is absent, `raw_snps_packed` uint8 [M_c x ceil(N*bits/8)] with `packed_bits` (1 or 2) and `num_indivs`
    (chunkstore.write_genotype_container(packed_bits=...)): 8x / 4x fewer bytes off the disk and over PCIe, expanded
    on the device (mmg_geno_upload_packed)."""
    if 'raw_snps' in cg:
        return cg['raw_snps'], 0
    return cg['raw_snps_packed'], int(np.asarray(cg['packed_bits'][...]))


def _num_indivs(cg):
    raw, bits = _raw_dataset(cg)
    return int(np.asarray(cg['num_indivs'][...])) if bits else int(np.asarray(raw[0:1]).shape[1])


def _chunk_plan(genot_data, min_maf, chunk_size, ramp=False):
    """[(chrom, kept row indices of the chunk, positions of the chunk)] without touching raw_snps.
    ramp: the first chunks of the stream are chunk_size / 8, / 4, / 2 -- the first upload is not hidden behind any
    compute, so a short first chunk starts the pipeline sooner while the later, full-size chunks keep the per-chunk costs
    of a scan (tail of the GEMM, the adaptive schedule's bookkeeping) amortised."""
    plan = []
    first = True
    for chrom in genot_data.keys():
        cg = genot_data[chrom]
        if min_maf is None:
            idx = np.arange(len(_raw_dataset(cg)[0]))
        else:
            idx = np.nonzero(_maf_filter(cg, min_maf))[0]
        positions = np.asarray(cg['positions'][...])[idx] if 'positions' in cg else idx
        i = 0
        if ramp and first:
            for div in (8, 4, 2):
                step = max(256, chunk_size // div // 256 * 256)
                if i + step >= len(idx):
                    break
                plan.append((chrom, idx[i:i + step], positions[i:i + step]))
                i += step
            first = False
        for j in range(i, len(idx), chunk_size):
            plan.append((chrom, idx[j:j + chunk_size], positions[j:j + chunk_size]))
    return plan


def _read_chunk(genot_data, chrom, sel, out=None):
    """Rows `sel` of a chromosome's raw_snps as a C-contiguous int8 block.  out: a reusable (page-locked) host buffer
    of at least len(sel) * N bytes -- a memory-mapped dataset is then read with one readinto() into it instead of
    being faulted in page by page through the mapping (2.8 GB/s at config 5's chunk size) and staged a second time."""
    raw, bits = _raw_dataset(genot_data[chrom])
    lo, hi = int(sel[0]), int(sel[-1]) + 1
    n_ind = raw.shape[1]                                                 # bytes per row (packed: ceil(N*bits/8))
    want = np.uint8 if bits else np.int8
    if (out is not None and isinstance(raw, np.memmap) and raw.dtype == want and raw.flags['C_CONTIGUOUS']
            and out.nbytes >= len(sel) * n_ind and getattr(raw, 'filename', None) is not None):
        base = raw.offset + lo * n_ind
        whole = out.nbytes >= (hi - lo) * n_ind                          # the span fits: one contiguous read, filtered in place
        rows_in = hi - lo if whole else len(sel)
        block = out.view(want)[:rows_in * n_ind].reshape(rows_in, n_ind)
        mv = memoryview(block).cast('B')

        def part(a, b, src=None):                                        # file bytes [src, src + b - a) -> buffer bytes [a, b)
            with open(raw.filename, 'rb', buffering=0) as f:             # (GIL released in readinto)
                f.seek(base + (a if src is None else src))
                got = a
                while got < b:
                    k = f.readinto(mv[got:b])
                    if not k:
                        raise IOError("short read from %s" % raw.filename)
                    got += k

        rel_idx = np.asarray(sel, dtype=np.int64) - lo
        if len(sel) != hi - lo:
            breaks = np.nonzero(np.diff(rel_idx) != 1)[0] + 1
            starts = np.concatenate(([0], breaks)).tolist()
            ends = np.concatenate((breaks, [len(rel_idx)])).tolist()
        nbytes = len(mv)
        nthr = min(_READ_THREADS, max(1, nbytes >> 23))                  # one reader per 8 MB, a page-cache copy each
        if not whole:
            # a MAF filter that keeps few rows stretches the span far beyond the chunk (chunk_size / kept fraction): the
            # page-locked buffers are bounded (_resident_chunks), so the kept runs are read one by one, each straight to
            # its place -- no row that the filter drops is read at all (advisor r4: 2 x 60 GB pinned at N = 50,000, 10 % kept)
            def runs(r0, r1):
                with open(raw.filename, 'rb', buffering=0) as f:
                    for s0, e0 in zip(starts[r0:r1], ends[r0:r1]):
                        f.seek(base + int(rel_idx[s0]) * n_ind)
                        got, end = s0 * n_ind, e0 * n_ind
                        while got < end:
                            k = f.readinto(mv[got:end])
                            if not k:
                                raise IOError("short read from %s" % raw.filename)
                            got += k
            if nthr == 1 or len(starts) < 2 * nthr:
                runs(0, len(starts))
            else:
                from concurrent.futures import ThreadPoolExecutor
                cuts = [len(starts) * t // nthr for t in range(nthr + 1)]
                with ThreadPoolExecutor(max_workers=nthr) as ex:
                    list(ex.map(lambda ab: runs(*ab), zip(cuts[:-1], cuts[1:])))
            return block
        if nthr == 1:
            part(0, nbytes)
        else:
            from concurrent.futures import ThreadPoolExecutor
            cuts = [nbytes * t // nthr // 4096 * 4096 for t in range(nthr)] + [nbytes]
            with ThreadPoolExecutor(max_workers=nthr) as ex:
                list(ex.map(lambda ab: part(*ab), zip(cuts[:-1], cuts[1:])))
        if len(sel) != hi - lo:
            # a MAF filter dropped rows: the kept ones move up inside the buffer, run by run (sel is ascending, so a row only
            # ever moves towards the front: memmove).  Before: the span was faulted in through the mapping and the subset
            # copied out by fancy indexing -- 5 GB/s of an int8 container against 50 through the page-locked buffer
            import ctypes
            addr = block.ctypes.data
            dst = 0
            for s0, e0 in zip(starts, ends):
                src, cnt = int(rel_idx[s0]), e0 - s0
                if src != dst:
                    ctypes.memmove(addr + dst * n_ind, addr + src * n_ind, cnt * n_ind)
                dst += cnt
            block = block[:len(sel)]
        return block
    block = np.asarray(raw[lo:hi])                                       # one contiguous read ...
    if len(sel) != hi - lo:
        block = block[sel - lo]                                          # ... then the MAF subset
    return np.ascontiguousarray(block, dtype=want)


def _upload_chunk(ctx, genot_data, chrom, block, store=None):
    """The chunk's rows into a (new or pooled) store: int8 rows as they are, packed rows through the device unpack."""
    cg = genot_data[chrom]
    _raw, bits = _raw_dataset(cg)
    if not bits:
        return ctx.geno(block) if store is None else store.reset(len(block)).upload(block)
    n = _num_indivs(cg)
    g = ctx.geno(M=len(block), N=n) if store is None else store.reset(len(block))
    return g.upload_packed(block, bits=bits)


class _Borrowed(object):
    """A pooled store handed to the chunk loop: close() gives it back instead of freeing HBM."""

    def __init__(self, g):
        self._g = g

    def __getattr__(self, name):
        return getattr(self._g, name)

    def close(self):
        pass


def _resident_chunks(ctx, genot_data, plan, rank=0, world=1, prefetch=True, reuse=False):
    """Yield (chunk index, chrom, Geno) for the chunks this rank owns (ci % world == rank).  With prefetch the NEXT
    owned chunk is read from the source and uploaded by a helper thread on a second context/stream of the same
    device while the caller computes on the current one (the C ABI is blocking; ctypes releases the GIL).
    reuse: the chunks ping-pong between TWO stores allocated once (mmg_geno_reset; hipMalloc / hipFree synchronise
    the device and would serialise the two streams) -- a yielded store is valid until the next-but-one chunk."""
    mine = [ci for ci in range(len(plan)) if ci % world == rank]
    if not (prefetch and isinstance(ctx, _lib.Context)):
        for ci in mine:
            chrom, sel, _pos = plan[ci]
            yield ci, chrom, _upload_chunk(ctx, genot_data, chrom, _read_chunk(genot_data, chrom, sel))
        return
    from concurrent.futures import ThreadPoolExecutor
    up = _UPLOAD_CTX.get(ctx.device)
    if up is None:
        up = _UPLOAD_CTX[ctx.device] = _lib.Context(ctx.device)         # second stream of the same device, kept
    pool, host = [], [None, None]
    if reuse and mine:
        cap = max(len(plan[ci][1]) for ci in mine)
        span = max(int(plan[ci][1][-1]) + 1 - int(plan[ci][1][0]) for ci in mine if len(plan[ci][1]))   # rows read per chunk
        cg0 = genot_data[plan[mine[0]][0]]
        n_ind = _num_indivs(cg0)
        # host bytes per SNP: N, or ceil(N bits / 8) when packed -- the LARGEST over the plan's chromosomes, and every
        # chromosome must hold the same individuals (a tree mixing raw_snps and raw_snps_packed chromosomes used to fail
        # inside _read_chunk's reshape in the middle of the stream; advisor r3)
        row_bytes = 0
        for chrom in dict.fromkeys(plan[ci][0] for ci in mine):
            cg = genot_data[chrom]
            if _num_indivs(cg) != n_ind:
                raise ValueError("chromosome %r holds %d individuals, %r holds %d: one stream needs one set of individuals"
                                 % (chrom, _num_indivs(cg), plan[mine[0]][0], n_ind))
            row_bytes = max(row_bytes, int(_raw_dataset(cg)[0].shape[1]))
        # two HBM stores + two page-locked staging buffers, kept between calls (kinship pass, scan pass, the next
        # file ...): allocating them costs ~0.1 s, as much as streaming 5 GB
        key = (ctx.device, n_ind, row_bytes)
        cached = _POOLS.get(key)
        # the staging buffers hold the SPAN of a chunk (first to last selected row: read contiguously, filtered in place) as long
        # as that is no more than the chunk's own rows or KIN_MAX_BYTES, whichever is larger; a sparser selection is read run by
        # run (_read_chunk), so a MAF filter cannot stretch the page-locked memory beyond 2 x max(chunk, 6 GB) (advisor r4)
        stage_rows = max(cap, min(span, int(KIN_MAX_BYTES) // max(1, row_bytes)))
        if cached is None or cached[0] < cap or cached[2][0].nbytes < stage_rows * row_bytes:
            if cached is not None:
                for g in cached[1]:
                    g.close()
            cached = _POOLS[key] = (cap, [up.geno(M=cap, N=n_ind) for _ in range(2)],
                                    [up.pinned_empty(stage_rows * row_bytes, dtype=np.int8) for _ in range(2)])
        pool, host = cached[1], cached[2]

    def load(ci, slot):
        chrom, sel, _pos = plan[ci]
        block = _read_chunk(genot_data, chrom, sel, out=host[slot])
        if pool:
            return ci, chrom, _Borrowed(_upload_chunk(up, genot_data, chrom, block, store=pool[slot]))
        return ci, chrom, _upload_chunk(up, genot_data, chrom, block)

    try:
        with ThreadPoolExecutor(max_workers=1) as ex:
            fut = ex.submit(load, mine[0], 0) if mine else None
            for k in range(len(mine)):
                cur = fut.result()
                fut = ex.submit(load, mine[k + 1], (k + 1) & 1) if k + 1 < len(mine) else None
                yield cur
    finally:
        pass                                                             # pooled stores stay allocated: release_pools()


def _dev_comm(coll):
    return getattr(coll, 'device_comm', None) if coll is not None else None


KINSHIP_ON_DEVICE_MIN_N = 8192   # run_emmax: from here the kinship goes accumulator -> REML workspace in HBM (see there; below,
                                 # the host round trip is milliseconds and keeps results bit-identical to passing k=)
KIN_MIN_ROWS = 65536       # SNPs per exact-GRM call from which the weights take 4 digit planes instead of 5 (api.hip)
KIN_MAX_BYTES = 6e9        # ... as long as one chunk's int8 rows stay below this (two stores + two staging buffers)


def _merge_plan(plan, n_indivs, min_rows=None, max_bytes=None):
    """Neighbouring chunks of one chromosome joined until each has >= min_rows SNPs: the kinship sum does not depend
    on how its SNPs are grouped, and mmg_kin_acc_add_grm is cheaper per SNP on larger groups (a fifth fewer digit-plane
    GEMMs from 65,536 SNPs on, one N x N combine pass per call: 388 -> 272 ms per 50,000 SNPs at N = 50,000)."""
    min_rows = KIN_MIN_ROWS if min_rows is None else min_rows
    max_rows = max(1, int((KIN_MAX_BYTES if max_bytes is None else max_bytes) // max(1, n_indivs)))
    out = []
    for chrom, sel, pos in plan:
        if (out and out[-1][0] == chrom and len(out[-1][1]) < min_rows and len(out[-1][1]) + len(sel) <= max_rows):
            out[-1] = (chrom, np.concatenate([out[-1][1], sel]), np.concatenate([out[-1][2], pos]))
        else:
            out.append((chrom, sel, pos))
    return out


def _ibd_kinship(ctx, genot_data, n_indivs, plan, coll=None, prefetch=True, timings=None, keep_device=False):
    """keep_device: return (_lib.DeviceKinship, n) -- the scaled kinship stays in HBM for the likelihood search
    (mmg_reml_create_from_acc); .host() downloads it when the caller wants the array."""
    rank, world = (coll.rank, coll.world) if coll is not None else (0, 1)
    acc = ctx.kinship_accumulator(n_indivs)
    merged = _merge_plan(plan, n_indivs)
    if len(merged) >= 2 * world or world == 1:                           # else keep every rank busy with the finer plan
        plan = merged
    for ci, chrom, g in _resident_chunks(ctx, genot_data, plan, rank, world, prefetch, reuse=True):
        acc.add_grm(g)                                                   # :99-106; a SNP with std == 0 is an error
        if timings is not None and isinstance(ctx, _lib.Context):
            timings['grm_kernel_s'] = timings.get('grm_kernel_s', 0.0) + 1e-3 * ctx.kernel_ms("grm")
        g.close()
    if coll is not None and world > 1:
        acc.allreduce(_dev_comm(coll))                                   # N x N partial sums never leave HBM
    n_all = sum(len(sel) for _c, sel, _p in plan)
    if hasattr(acc, 'scale_k'):
        # :107-111 on the device: scale_k(K / n) = scale_k(K) (the rule is homogeneous of degree 0 in K), so the sum is
        # scaled as it lies in HBM and crosses PCIe once -- three host passes over 20 GB less at N = 50,000
        acc.scale_k()
        if keep_device and isinstance(ctx, _lib.Context):
            assert acc.snps() == n_all, (acc.snps(), n_all)
            # (a helper thread downloading the 20 GB on a third context's stream while the reduction runs was tried:
            # the download's host side -- first touch of 20 GB, the runtime's staging copies -- holds the launches of the
            # 12,000-kernel band reduction back: REML 6.0 -> 6.6 s, scan model 2.4 -> 3.6 s on the share-of-8 run)
            return _lib.DeviceKinship(acc, scaled=True), n_all
        k_mat, n_snps = acc.fetch()
        acc.close()
        assert n_snps == n_all, (n_snps, n_all)                          # after the all-reduce: every rank's chunks counted
        return k_mat, n_all
    k_mat, n_snps = acc.fetch()
    acc.close()
    assert n_snps == n_all, (n_snps, n_all)
    k_mat = k_mat / float(n_all)                                         # :107
    return kinship.scale_k(k_mat), n_all                                 # :108-111 (inline scale_k)


def _ibd_kinship_normalised(ctx, genot_data, n_indivs, chunk_size, coll=None):
    """hdf5_data.py:37-44 for trees that carry pre-normalised float `snps` datasets (a chromosome without one is
    standardised per SNP as the reference does it, :40-43): K = sum_chunks x'x / n_snps over the rows x of each chunk --
    a device fp64 GEMM per chunk (mmg_dgemm_f64: the values are arbitrary floats, so the exact int8 routes do not apply),
    chunks dealt round-robin over the ranks and the partial sums all-reduced; then scale_k's rule (:46-49)."""
    rank, world = (coll.rank, coll.world) if coll is not None else (0, 1)
    k_sum = np.zeros((n_indivs, n_indivs))
    n_snps, ci = 0, 0
    for chrom in genot_data.keys():
        cg = genot_data[chrom]
        normalised = 'snps' in cg.keys()
        ds = cg['snps'] if normalised else cg['raw_snps']
        num = len(ds)
        for i in range(0, num, chunk_size):
            end = min(i + chunk_size, num)
            n_snps += end - i
            mine = ci % world == rank
            ci += 1
            if not mine:
                continue
            x = np.asarray(ds[i:end], dtype=np.float64)
            if not normalised:
                sd = x.std(1)
                if np.any(sd == 0):
                    raise ValueError("a SNP without variation cannot be standardised (std == 0)")
                x = (x - x.mean(1, keepdims=True)) / sd[:, None]
            k_sum += ctx.dgemm(x, x, ta=True)                            # x'x: [N x rows] [rows x N]
    if coll is not None and world > 1:
        k_sum = np.asarray(coll.allreduce(k_sum, "sum")).reshape(n_indivs, n_indivs)
    return kinship.scale_k(k_sum / float(n_snps)), n_snps


def _warn_maf_ignored(min_maf):
    if min_maf is not None:
        import warnings
        warnings.warn("calculate_ibd_kinship: min_maf is ignored for trees with pre-normalised `snps` datasets (the "
                      "reference's own function has no filter, hdf5_data.py:17-62)")


def calculate_ibd_kinship(hdf5_filename, n_indivs=None, min_maf=None, chunk_size=100000, overwrite=False, ctx=None,
                          coll=None):
    """hdf5_data.py:17-62: K = sum_m z_m z_m' / n_snps with z = (s - mean)/std per SNP, scaled with scale_k's rule.
    The sum lives in HBM across chunks (mmg_kin_acc_*).  Given a file name the kinship is stored in the file as the
    'kinship' dataset (:60) unless it is already there (overwrite=False); given a genot_data tree it is returned
    as (K, n_snps).  min_maf=None: no MAF filter (the reference's stand-alone function has none; its run_emmax twin
    filters, :91-96)."""
    ctx = ctx or _lib.get_context()
    if isinstance(hdf5_filename, str):
        h5f = chunkstore.open_container(hdf5_filename, 'a')
        n = len(h5f['indiv_data']['indiv_ids'][...])
        if 'kinship' in h5f.keys() and not overwrite:
            return np.asarray(h5f['kinship'][...]), None
        if any('snps' in h5f['genot_data'][chrom].keys() for chrom in h5f['genot_data'].keys()):
            _warn_maf_ignored(min_maf)
            k, n_snps = _ibd_kinship_normalised(ctx, h5f['genot_data'], n, chunk_size, coll)   # :37,44
        else:
            plan = _chunk_plan(h5f['genot_data'], min_maf, chunk_size)
            k, n_snps = _ibd_kinship(ctx, h5f['genot_data'], n, plan, coll)
        if coll is None or coll.rank == 0:
            if 'kinship' in h5f.keys():
                del h5f['kinship']
            h5f.create_dataset('kinship', data=k)
            h5f.flush()
        return k, n_snps
    if any('snps' in hdf5_filename[chrom].keys() for chrom in hdf5_filename.keys()):
        _warn_maf_ignored(min_maf)
        return _ibd_kinship_normalised(ctx, hdf5_filename, n_indivs, chunk_size, coll)
    plan = _chunk_plan(hdf5_filename, min_maf, chunk_size)
    return _ibd_kinship(ctx, hdf5_filename, n_indivs, plan, coll)


def _gather_owned(parts, plan, coll):
    """parts {chunk index: values of an owned chunk} on every rank -> {chunk index: values} for ALL chunks:
    one all-gather of each rank's owned values (padded to the largest share)."""
    world, rank = coll.world, coll.rank
    share = [sum(len(plan[ci][1]) for ci in range(len(plan)) if ci % world == r) for r in range(world)]
    count = max(share) if share else 0
    if count == 0:
        return dict(parts)
    mine = np.full(count, np.nan)
    if parts:
        flat = np.concatenate([parts[ci] for ci in sorted(parts)])
        mine[:len(flat)] = flat
    gathered = np.asarray(coll.allgather(mine)).reshape(world, count)
    out, cursor = {}, [0] * world
    for ci in range(len(plan)):
        r = ci % world
        n = len(plan[ci][1])
        out[ci] = gathered[r, cursor[r]:cursor[r] + n]
        cursor[r] += n
    return out


def run_emmax(hdf5_filename, out_file=None, min_maf=0.1, recalculate_kinship=True, chunk_size=100000, k=None,
              ctx=None, coll=None, num_perm=0, perm_idx=None, phenotypes=None, prefetch=True, fast_perm=True,
              eigen_free=None, timings=None, perm_h=None):
    """hdf5_data.py:70-187 (and :191-351 when num_perm > 0).  fast_perm: the permutation test of a chunk reuses the
    quadratic forms of the scan that just ran over it (mmg_emmax_perm_after_scan) instead of recomputing them.
    eigen_free: REML and the scan model from Cholesky factorisations instead of eigh(K) (linear_models.
    get_estimates_eigen_free); default: when N > linear_models.EIGEN_FREE_MIN_N (mandatory beyond rocSOLVER's syevd range,
    N > 46,340) -- since round 5 with a permutation test too, whose H_sqrt_inv is then L^-1 of K + delta I = L L'
    (linear_models.perm_h_from_cholesky; MMG_PERM_H=eigen: the eigendecomposition's matrix).

    hdf5_filename: container path (chunkstore / HDF5) or an open genot_data tree / mapping.  For the reference's
    call shape `run_emmax(genot_data, phenotypes, ...)` of round 1 the second positional argument may be the
    phenotype vector.  out_file: result container to write (:142-184: pseudo_heritability, ve, vg, max_ll,
    num_snps, chrom_results/<chrom>/{ps, positions}; perm: kinship, perm_min_ps, perm_max_f_stats, five_perc_*).
    Returns {'pseudo_heritability','ve','vg','max_ll','num_snps','kinship','chrom_results': {chrom: {'ps',
    'positions'}}} plus 'perm_min_ps', 'perm_max_f_stats', 'threshold_05' for the permutation variant.

    Footprint of the streaming pools (kept between calls until release_pools()): two HBM chunk stores and two page-locked
    host staging buffers of the largest chunk each -- chunk_size SNPs for the scan pass, and up to KIN_MAX_BYTES (6 GB) of
    int8 rows per buffer for the kinship pass, whose chunks are merged to >= 65,536 SNPs (_merge_plan): at most 2 x 6 GB
    of HBM, and 2 x max(largest chunk, 6 GB) of pinned host memory whatever the MAF filter keeps (a chunk whose span of
    file rows exceeds that is read run by run: _read_chunk).  All chromosomes of a stream must hold the same individuals; the buffers
    are sized by the largest row (raw int8 or bit-packed) over the plan.

    perm_idx / perm_h: replay of a recorded permutation test -- the [num_perm x N] shuffles and the H_sqrt_inv they were
    applied in.  A permutation's outcome depends on the row signs of H_sqrt_inv (the ROTATED residuals are what
    linear_models.py:1151-1154 shuffles) and those are the eigensolver's choice, so reproducing the numbers of a reference
    run takes the reference's matrix; without perm_h the test runs in this model's own H (a draw from the same null).

    timings: a dict that receives the wall seconds of the stages (kinship_pass_s, reml_s, scan_model_s, scan_pass_s,
    gather_s) and the route taken -- what bench.py --mode c5 reports per rank."""
    ctx = ctx or _lib.get_context()
    _t = [time.time()]

    def _lap(key):
        if timings is not None:
            now = time.time()
            timings[key] = timings.get(key, 0.0) + now - _t[0]
            _t[0] = now
    if out_file is not None and not isinstance(out_file, str):           # run_emmax(genot_data, phenotypes, ...)
        phenotypes, out_file = out_file, None
    ih5f = None
    if isinstance(hdf5_filename, str):
        ih5f = chunkstore.open_container(hdf5_filename, 'r')
        genot_data = ih5f['genot_data']
        if phenotypes is None:
            phenotypes = ih5f['indiv_data']['phenotypes'][...]           # :119
        if not recalculate_kinship and k is None:
            assert 'kinship' in ih5f.keys(), 'Kinship is missing.  Please calculate that first!'   # :114
            k = np.asarray(ih5f['kinship'][...])
    else:
        genot_data = hdf5_filename
    phenotypes = np.asarray(phenotypes, dtype=np.float64).reshape(-1)
    n = len(phenotypes)
    rank, world = (coll.rank, coll.world) if coll is not None else (0, 1)
    plan = _chunk_plan(genot_data, min_maf, chunk_size)
    if k is None:
        # the kinship goes from the accumulator into the likelihood search without visiting the host (and comes down in the
        # background for the result) when the route that follows is the device's: at N = 50,000 the download, the host's
        # second scale_k (:121 -> linear_models.py:580) and the upload, 20 GB each, were ~2 s of the REML stage
        on_device = (os.environ.get('MMG_KINSHIP_ON_DEVICE', '1') != '0' and isinstance(ctx, _lib.Context)
                     and (not num_perm or lm.perm_h_from_cholesky(ctx)) and eigen_free is not False
                     and n >= KINSHIP_ON_DEVICE_MIN_N and n > lm.EIGEN_FREE_MIN_N)
        k, n_snps = _ibd_kinship(ctx, genot_data, n, plan, coll, prefetch, timings, keep_device=on_device)
    else:
        n_snps = sum(len(sel) for _c, sel, _p in plan)
    _lap('kinship_pass_s')
    lmm = lm.LinearMixedModel(phenotypes, ctx=ctx)                       # :121
    lmm.add_random_effect(k)
    # Round 5: the permutation test no longer forces the eigendecomposition -- its H_sqrt_inv is L^-1 of K + delta I = L L' as
    # the REML workspace holds it (linear_models.perm_h_from_cholesky; MMG_PERM_H=eigen or eigen_free=False: the literal
    # route through eigh(K), 265 ms of rocSOLVER at N = 5000 on every rank)
    perm_chol = bool(num_perm) and lm.perm_h_from_cholesky(ctx)
    if eigen_free and num_perm and not perm_chol:
        raise ValueError("run_emmax: MMG_PERM_H=eigen asks for the eigendecomposition's H_sqrt_inv, which the "
                         "eigendecomposition-free route does not produce -- call with eigen_free=False (or None)")
    if eigen_free is None:
        eigen_free = n > lm.EIGEN_FREE_MIN_N and (not num_perm or perm_chol) and hasattr(ctx, 'reml')
    res = lmm._try_eigen_free(coll=coll) if eigen_free else None         # :126-137 without either eigendecomposition
    reml_ws = None
    if res is not None:
        _lap('reml_s')
        if timings is not None:
            timings['route'] = 'eigendecomposition-free (REML through %s)' % (
                'one band reduction of K' if res['reml'].uses_band() else 'a Cholesky factorisation per delta')
        prep = lmm.scan_model_eigen_free(res)
        reml_ws = res.pop('reml')
        if not num_perm:
            reml_ws.close()
            reml_ws = None
    else:
        eig_L = lmm._get_eigen_L_()                                      # :126
        res = lmm.get_estimates(eig_L, method='REML')                    # :131-137 (no eig_R: linear_models._SpectralSumsL)
        _lap('reml_s')
        if timings is not None:
            timings['route'] = 'eigh'
        prep = lmm.scan_prepare(res['H_sqrt_inv'])
        ctx.scan_set_model(prep['A'], prep['w'], 0)
    _lap('scan_model_s')
    out = {'pseudo_heritability': res['pseudo_heritability'], 've': res['ve'], 'vg': res['vg'],
           'max_ll': res['max_ll'], 'num_snps': n_snps, 'chrom_results': {}, 'kinship': k}
    k_dev = k if isinstance(k, _lib.DeviceKinship) else None
    chroms = list(genot_data.keys())
    parts = {}
    pp = None
    if num_perm:                                                         # :262-330: SNP-independent part, once
        lmm_p = lm.LinearMixedModel(phenotypes, ctx=ctx)                 # perm_prepare centres Y in place
        lmm_p.add_random_effect(k)
        if perm_h is not None:
            if reml_ws is not None:
                reml_ws.close()
                reml_ws = None
            if 'HtQ' not in prep:                                            # A = H'H - (H'Q)(H'Q)' does not depend on the signs
                from scipy import linalg as _la
                Hh = np.asarray(perm_h, dtype=np.float64)
                prep['HtQ'] = np.ascontiguousarray((Hh.T @ _la.qr(Hh @ lmm.X, mode='economic')[0]).T)
            pp = lmm_p.perm_prepare(perm_h, num_perm=num_perm, perm_idx=perm_idx)
            plan_p = ctx.perm_plan(pp['H'], pp['Ys'], pp['h0_rss'])
        elif reml_ws is not None:
            try:
                # the scan model has left L^-1 of this delta in the workspace: H X, H y are two triangular products, the
                # plan's operand images are built from the matrix where it lies, and the rows H'Q_c of the after-scan
                # form (A = H'H - sum_c u_c u_c') are L^-T Q
                pp = lmm_p.perm_prepare(None, num_perm=num_perm, perm_idx=perm_idx, reml=reml_ws, delta=res['delta'])
                plan_p = reml_ws.perm_plan(res['delta'], pp['Ys'], pp['h0_rss'])
                from scipy import linalg as _la
                Qc = _la.qr(reml_ws.linv_apply(res['delta'], lmm.X), mode='economic')[0]
                prep['HtQ'] = np.ascontiguousarray(reml_ws.linv_apply(res['delta'], Qc, trans=True).T)
            finally:
                reml_ws.close()
        else:
            pp = lmm_p.perm_prepare(res['H_sqrt_inv'], num_perm=num_perm, perm_idx=perm_idx)
            plan_p = ctx.perm_plan(pp['H'], pp['Ys'], pp['h0_rss'])      # operand images of the test, once for all chunks
        min_rss = np.full(num_perm, pp['h0_rss'])
    for ci, chrom, g in _resident_chunks(ctx, genot_data, plan, rank, world, prefetch, reuse=True):
        parts[ci] = ctx.scan(g, prep['h0_rss'], prep['n_p'])['ps']       # :174 _emmax_f_test_(emma_num=0)
        if timings is not None and isinstance(ctx, _lib.Context):
            timings['scan_kernel_s'] = timings.get('scan_kernel_s', 0.0) + 1e-3 * ctx.kernel_ms("scan_quad")
        # :294-311,330 -- the permutation test runs on every chromosome but the LAST (`chr12_snps`); here chunk by
        # chunk right behind the scan of the same chunk, whose quadratic forms it reuses (same H; t.t needs only
        # 1 + q dot products per SNP on top of them)
        if num_perm and chrom != chroms[-1]:
            mr = plan_p.run(g, after_scan_HtQ=prep['HtQ'] if (fast_perm and isinstance(ctx, _lib.Context)) else None)
            min_rss = np.minimum(min_rss, mr)
        g.close()
    _lap('scan_pass_s')
    if coll is not None and world > 1:
        parts = _gather_owned(parts, plan, coll)                         # every SNP was scanned by exactly one rank
    _lap('gather_s')
    for ci, (chrom, _sel, pos) in enumerate(plan):
        d = out['chrom_results'].setdefault(chrom, {'ps': [], 'positions': []})
        d['ps'].append(parts[ci])
        d['positions'].append(pos)
    for chrom in chroms:
        d = out['chrom_results'].setdefault(chrom, {'ps': [np.zeros(0)], 'positions': [np.zeros(0, dtype=np.int64)]})
        d['ps'] = np.concatenate(d['ps'])
        d['positions'] = np.concatenate(d['positions'])
    out['chrom_results'] = {chrom: out['chrom_results'][chrom] for chrom in chroms}      # the file's chromosome order
    if num_perm:                                                         # :339-347
        plan_p.close()
        if coll is not None and world > 1:
            min_rss = coll.allreduce(min_rss, "min")
        max_f = (pp['h0_rss'] / min_rss - 1.0) * pp['n_p']               # linear_models.py:1171
        min_ps = ctx.f_sf(max_f, pp['n_p'])                              # :1172
        order = np.argsort(min_ps)
        five = num_perm // 20                                            # :342
        out.update(perm_min_ps=min_ps, perm_max_f_stats=max_f,
                   threshold_05=(float(min_ps[order][five]), float(max_f[order][five])))
    if k_dev is not None:                                                # the result carries the array, as always
        out['kinship'] = k_dev.host()
        k_dev.close()
    if out_file is not None and rank == 0:
        _write_results(out_file, out, ih5f, num_perm)
    return out


def _write_results(out_file, out, ih5f, num_perm):
    """The result file of hdf5_data.py:142-184 (+ :241-243,339-347 for the permutation variant)."""
    oh5f = chunkstore.open_container(out_file, 'a')
    oh5f.create_dataset('pseudo_heritability', data=np.array(out['pseudo_heritability']))
    oh5f.create_dataset('ve', data=np.array(out['ve']))
    oh5f.create_dataset('vg', data=np.array(out['vg']))
    oh5f.create_dataset('max_ll', data=np.array(out['max_ll']))
    # :150 copies the INPUT file's num_snps (all SNPs, before the MAF filter); the permutation driver stores `n_snps` (:289),
    # a name its counting loop has reused (:253): the kept count of the LAST chromosome (pinned by tests/golden/hdf5_n200.npz)
    if num_perm:
        n_in = np.array(len(list(out['chrom_results'].values())[-1]['ps']) if out['chrom_results'] else 0)
    elif ih5f is not None and 'num_snps' in ih5f.keys():
        n_in = np.array(ih5f['num_snps'][...])
    else:
        n_in = np.array(out['num_snps'])
    oh5f.create_dataset('num_snps', data=n_in)
    crg = oh5f.create_group('chrom_results')
    for chrom, d in out['chrom_results'].items():
        g = crg.create_group(str(chrom))
        g.create_dataset('ps', data=d['ps'])
        g.create_dataset('positions', data=d['positions'])
    if num_perm:
        oh5f.create_dataset('kinship', data=out['kinship'])              # :242
        # :339-347 -- both arrays are sorted ASCENDING (the `[::-1]` at :341 is a no-op expression), so the stored
        # five_perc_perm_max_f_stats is the 5 % point from the BOTTOM of the max-F distribution; kept as is for
        # identical files.  (`threshold_05` in the returned dict pairs the 5 % p-value with its own statistic.)
        ps_sorted = np.sort(out['perm_min_ps'])
        f_sorted = np.sort(out['perm_max_f_stats'])
        five = num_perm // 20
        oh5f.create_dataset('perm_min_ps', data=ps_sorted)
        oh5f.create_dataset('perm_max_f_stats', data=f_sorted)
        oh5f.create_dataset('five_perc_perm_min_ps', data=np.array(ps_sorted[five]))
        oh5f.create_dataset('five_perc_perm_max_f_stats', data=np.array(f_sorted[five]))
    oh5f.flush()
    oh5f.close()


def run_emmax_perm(hdf5_filename, out_file=None, min_maf=0.1, recalculate_kinship=True, chunk_size=100000,
                   num_perm=500, perm_idx=None, k=None, ctx=None, coll=None, phenotypes=None, prefetch=True,
                   fast_perm=True, perm_h=None):
    """hdf5_data.py:191-351 (the reference always recalculates the kinship here; pass k to skip that)."""
    return run_emmax(hdf5_filename, out_file, min_maf=min_maf, chunk_size=chunk_size, k=k, ctx=ctx, coll=coll,
                     num_perm=num_perm, perm_idx=perm_idx, phenotypes=phenotypes, prefetch=prefetch,
                     fast_perm=fast_perm, perm_h=perm_h)


def run_emmax_multi(hdf5_filename, out_file=None, phenotypes=None, min_maf=0.1, chunk_size=50000, k=None, ctx=None,
                    coll=None, prefetch=True):
    """Several phenotypes over one streamed genotype file: what a user of the reference gets by calling run_emmax
    (:70-187) once per phenotype, computed in ONE pass over the SNPs after the kinship -- per chunk the SNPs are
    rotated into the eigenbasis of K (mmg_rot_load) and every phenotype, with its own delta and null model, is an
    HBM-bound pass over the rotated chunk (mmg_emmax_scan_multi; linear_models.emmax_multi is the in-memory form).
    phenotypes: [P x N] (default: the file's indiv_data/phenotypes, 1-D or 2-D).  Multi-GPU: chunks round-robin,
    owned blocks all-gathered per phenotype.  Result file: the scalars of run_emmax as length-P arrays and
    chrom_results/<chrom>/{ps [P x M_c], positions}.  N must be within the eigensolver's range (N <= 46,340)."""
    ctx = ctx or _lib.get_context()
    ih5f = None
    if isinstance(hdf5_filename, str):
        ih5f = chunkstore.open_container(hdf5_filename, 'r')
        genot_data = ih5f['genot_data']
        if phenotypes is None:
            phenotypes = ih5f['indiv_data']['phenotypes'][...]
    else:
        genot_data = hdf5_filename
    ys = np.atleast_2d(np.asarray(phenotypes, dtype=np.float64))
    P, n = ys.shape
    rank, world = (coll.rank, coll.world) if coll is not None else (0, 1)
    plan = _chunk_plan(genot_data, min_maf, chunk_size)
    if k is None:
        k, n_snps = _ibd_kinship(ctx, genot_data, n, plan, coll, prefetch)
    else:
        n_snps = sum(len(sel) for _c, sel, _p in plan)
    lmm0 = lm.LinearMixedModel(ys[0], ctx=ctx)
    lmm0.add_random_effect(k)
    eig_L = lmm0._get_eigen_L_()
    models, d, omega, G = lm._multi_models(ys, lmm0.X, eig_L)
    h0 = np.array([m['h0_rss'] for m in models])
    n_p = n - (lmm0.X.shape[1] + 1)
    cap = max([len(sel) for _c, sel, _p in plan] or [1])
    rot = ctx.rot(eig_L['vectors'], cap)
    parts = {}
    try:
        for ci, chrom, g in _resident_chunks(ctx, genot_data, plan, rank, world, prefetch, reuse=True):
            rot.load(g)
            parts[ci] = ctx.scan_multi(rot, d, omega, G, h0, n_p, want=("ps",))["ps"]      # [P x rows]
            g.close()
    finally:
        rot.close()
    if coll is not None and world > 1:
        gathered = [_gather_owned({ci: v[p] for ci, v in parts.items()}, plan, coll) for p in range(P)]
        parts = {ci: np.vstack([gathered[p][ci] for p in range(P)]) for ci in range(len(plan))}
    out = {'num_snps': n_snps, 'kinship': k, 'chrom_results': {}}
    for key in ('pseudo_heritability', 've', 'vg', 'max_ll', 'delta'):
        out[key] = np.array([m[key] for m in models])
    for ci, (chrom, _sel, pos) in enumerate(plan):
        dct = out['chrom_results'].setdefault(chrom, {'ps': [], 'positions': []})
        dct['ps'].append(parts[ci])
        dct['positions'].append(pos)
    for chrom in genot_data.keys():
        dct = out['chrom_results'].setdefault(chrom, {'ps': [np.zeros((P, 0))], 'positions': [np.zeros(0, dtype=np.int64)]})
        dct['ps'] = np.concatenate(dct['ps'], axis=1)
        dct['positions'] = np.concatenate(dct['positions'])
    if out_file is not None and rank == 0:
        oh5f = chunkstore.open_container(out_file, 'a')
        for key in ('pseudo_heritability', 've', 'vg', 'max_ll'):
            oh5f.create_dataset(key, data=out[key])
        oh5f.create_dataset('num_snps', data=np.array(n_snps))
        crg = oh5f.create_group('chrom_results')
        for chrom, dct in out['chrom_results'].items():
            gq = crg.create_group(str(chrom))
            gq.create_dataset('ps', data=dct['ps'])
            gq.create_dataset('positions', data=dct['positions'])
        oh5f.flush()
        oh5f.close()
    return out

"""Chunked EMMAX drivers -- the compute of /root/reference/hdf5_data.py run_emmax (:70-187) and
run_emmax_perm (:191-351): MAF filter, GRM kinship accumulated over SNP chunks, REML once, EMMAX scan
per chromosome, optional permutation thresholds -- over a chunk source instead of an open HDF5 file.

Data source ("genot_data" of plink2hdf5.py:27-28,111-118): a mapping
    {chrom: {'raw_snps': int8 [M_c x N] (ndarray / memmap / h5py dataset), 'freqs': [M_c],
             'positions': [M_c]}}
Only `raw_snps[i:j]` slicing is used, so numpy memmaps and h5py datasets both work; `open_hdf5`
wraps a file in the reference's layout when h5py is installed (it is not in this image).

Multi-GPU (`coll`, mixmogam_amd.dist): chunks are dealt round-robin to ranks; the kinship partial sums
are all-reduced (SUM), eigh/REML are replicated, every rank scans its chunks and the per-chunk
p-values / permutation minima are combined on the host with all-reduces.
"""
import numpy as np

from . import _lib, kinship
from . import linear_models as lm


def open_hdf5(filename):
    """The reference's on-disk layout (plink2hdf5.py) as a chunk source.  Needs h5py."""
    try:
        import h5py
    except ImportError:
        raise ImportError("h5py is not installed; pass a mapping of arrays / memmaps instead")
    f = h5py.File(filename, 'r')
    return {'genot_data': f['genot_data'], 'phenotypes': f['indiv_data']['phenotypes'][...], 'file': f}


_UPLOAD_CTX = {}


def _maf_filter(cg, min_maf):
    freqs = np.asarray(cg['freqs'][...], dtype=np.float64)
    return np.minimum(freqs, 1 - freqs) > min_maf                       # hdf5_data.py:91-93


def _chunks(genot_data, min_maf, chunk_size):
    """Yield (chrom, filtered chunk int8 [m x N], positions of the chunk)."""
    for chrom in genot_data.keys():
        cg = genot_data[chrom]
        keep = _maf_filter(cg, min_maf)
        idx = np.nonzero(keep)[0]
        positions = np.asarray(cg['positions'][...])[keep]
        raw = cg['raw_snps']
        for i in range(0, len(idx), chunk_size):
            sel = idx[i:i + chunk_size]
            lo, hi = int(sel[0]), int(sel[-1]) + 1
            block = np.asarray(raw[lo:hi])                               # one contiguous read ...
            if len(sel) != hi - lo:
                block = block[sel - lo]                                  # ... then the MAF subset
            yield chrom, np.ascontiguousarray(block, dtype=np.int8), positions[i:i + chunk_size]


def _resident_chunks(ctx, genot_data, min_maf, chunk_size, rank=0, world=1, prefetch=True):
    """Yield (chunk index, chrom, Geno or None, block rows, positions) for every chunk; chunks owned by
    other ranks come with Geno = None.  With prefetch the NEXT owned chunk is read from the source and
    uploaded by a helper thread on a second context/stream of the same device while the caller computes
    on the current one (the C ABI is blocking; ctypes releases the GIL): PCIe ingest overlaps the kernels."""
    items = enumerate(_chunks(genot_data, min_maf, chunk_size))
    if not (prefetch and isinstance(ctx, _lib.Context)):
        for ci, (chrom, block, pos) in items:
            yield ci, chrom, (ctx.geno(block) if ci % world == rank else None), len(block), pos
        return
    from concurrent.futures import ThreadPoolExecutor
    up = _UPLOAD_CTX.get(ctx.device)
    if up is None:
        up = _UPLOAD_CTX[ctx.device] = _lib.Context(ctx.device)     # second stream of the same device, kept

    def load_next():
        for ci, (chrom, block, pos) in items:
            return ci, chrom, (up.geno(block) if ci % world == rank else None), len(block), pos
        return None

    with ThreadPoolExecutor(max_workers=1) as pool:
        fut = pool.submit(load_next)
        while True:
            cur = fut.result()
            if cur is None:
                break
            fut = pool.submit(load_next)
            yield cur


def calculate_ibd_kinship(genot_data, n_indivs, min_maf=0.0, chunk_size=100000, ctx=None, coll=None):
    """hdf5_data.py:17-62 / :84-115: K = sum_m z_m z_m' / n_snps with z = (s - mean)/std per SNP, scaled
    with scale_k's rule.  The sum lives in HBM across chunks (mmg_kin_acc_*)."""
    ctx = ctx or _lib.get_context()
    acc = ctx.kinship_accumulator(n_indivs)
    n_snps = 0
    rank, world = (coll.rank, coll.world) if coll is not None else (0, 1)
    for ci, chrom, g, nrows, _pos in _resident_chunks(ctx, genot_data, min_maf, chunk_size, rank, world):
        n_snps += nrows
        if g is None:
            continue
        mean, sd = g.snp_stats()
        if np.any(sd == 0):
            raise ValueError("monomorphic SNP passed the MAF filter on chromosome %s" % chrom)
        acc.add(g, 1.0 / sd, -mean / sd)
        g.close()
    k_mat, _ = acc.fetch()
    acc.close()
    if coll is not None:
        k_mat = coll.allreduce(k_mat, "sum").reshape(n_indivs, n_indivs)
    k_mat = k_mat / float(n_snps)                                        # :107
    return kinship.scale_k(k_mat), n_snps                                # :108-111 (inline scale_k)


def run_emmax(genot_data, phenotypes, min_maf=0.1, chunk_size=100000, k=None, ctx=None, coll=None,
              num_perm=0, perm_idx=None):
    """hdf5_data.py:70-187 (and :191-351 when num_perm > 0).  Returns
    {'pseudo_heritability','ve','vg','max_ll','num_snps','chrom_results': {chrom: {'ps','positions'}}}
    plus 'perm_min_ps', 'perm_max_f_stats', 'threshold_05' for the permutation variant (:339-347)."""
    ctx = ctx or _lib.get_context()
    phenotypes = np.asarray(phenotypes, dtype=np.float64).reshape(-1)
    n = len(phenotypes)
    rank, world = (coll.rank, coll.world) if coll is not None else (0, 1)
    if k is None:
        k, n_snps = calculate_ibd_kinship(genot_data, n, min_maf, chunk_size, ctx, coll)
    else:
        n_snps = sum(int(_maf_filter(genot_data[c], min_maf).sum()) for c in genot_data.keys())
    lmm = lm.LinearMixedModel(phenotypes, ctx=ctx)                       # :121
    lmm.add_random_effect(k)
    eig_L = lmm._get_eigen_L_()                                          # :126
    res = lmm.get_estimates(eig_L, method='REML')                        # :131-137 (no eig_R: linear_models._SpectralSumsL)
    out = {'pseudo_heritability': res['pseudo_heritability'], 've': res['ve'], 'vg': res['vg'],
           'max_ll': res['max_ll'], 'num_snps': n_snps, 'chrom_results': {}, 'kinship': k}
    prep = lmm.scan_prepare(res['H_sqrt_inv'])
    ctx.scan_set_model(prep['A'], prep['w'], 0)
    per_chrom = {}
    kept = []                                                            # chunk genotype stores for the permutations
    for ci, chrom, g, nrows, pos in _resident_chunks(ctx, genot_data, min_maf, chunk_size, rank, world):
        ps = np.full(nrows, np.nan)
        if g is not None:
            ps = ctx.scan(g, prep['h0_rss'], prep['n_p'])['ps']          # :174 _emmax_f_test_(emma_num=0)
            if num_perm:
                kept.append(g)
            else:
                g.close()
        per_chrom.setdefault(chrom, []).append((ps, pos))
    for chrom, parts in per_chrom.items():
        ps = np.concatenate([p for p, _ in parts])
        if coll is not None:                                             # every SNP was scanned by exactly one rank
            ps = coll.allreduce(np.where(np.isnan(ps), np.inf, ps), "min")
        out['chrom_results'][chrom] = {'ps': ps, 'positions': np.concatenate([q for _, q in parts])}
    if num_perm:                                                         # :262-347
        lmm_p = lm.LinearMixedModel(phenotypes, ctx=ctx)
        lmm_p.add_random_effect(k)
        if perm_idx is None:
            idx = np.asmatrix(np.arange(n).reshape(n, 1))
            perm_idx = []
            for _ in range(num_perm):
                np.random.shuffle(idx)
                perm_idx.append(np.asarray(idx).reshape(-1).copy())
        min_ps, max_f = None, None
        for g in kept:
            lmm_c = lm.LinearMixedModel(phenotypes, ctx=ctx)             # _emmax_permutations_ centres Y in place
            lmm_c.add_random_effect(k)
            r = lmm_c._emmax_permutations_(g, k, res['H_sqrt_inv'], num_perm=num_perm, perm_idx=perm_idx)
            g.close()
            max_f = r['max_f_stats'] if max_f is None else np.maximum(max_f, r['max_f_stats'])
            min_ps = r['min_ps'] if min_ps is None else np.minimum(min_ps, r['min_ps'])
        if min_ps is None:
            min_ps, max_f = np.ones(num_perm), np.zeros(num_perm)
        if coll is not None:
            min_ps = coll.allreduce(min_ps, "min")
            max_f = coll.allreduce(max_f, "max")
        order = np.argsort(min_ps)
        out.update(perm_min_ps=min_ps, perm_max_f_stats=max_f,
                   threshold_05=(float(min_ps[order][num_perm // 20]), float(max_f[order][num_perm // 20])))
    return out


def run_emmax_perm(genot_data, phenotypes, min_maf=0.1, chunk_size=100000, num_perm=500, perm_idx=None, k=None,
                   ctx=None, coll=None):
    """hdf5_data.py:191-351."""
    return run_emmax(genot_data, phenotypes, min_maf=min_maf, chunk_size=chunk_size, k=k, ctx=ctx, coll=coll,
                     num_perm=num_perm, perm_idx=perm_idx)

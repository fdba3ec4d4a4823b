"""Kinship matrices -- same call surface as the reference's kinship.py, GEMMs on the MI355X.

calc_ibs_kinship / calc_ibd_kinship / scale_k / prepare_k follow /root/reference/kinship.py
(:14-56, :59-75, :94-100, :79-90).  Results are float64 ndarrays (the reference's IBS result is
a float64 numpy.matrix; ndarray supports the same indexing the callers use).
"""
import numpy as np

from . import _lib


def _as_snp_matrix(snps, dtype=np.int8):
    """list of M int8 arrays of length N, or 2-D array -> C-contiguous [M x N]."""
    return _lib.as_store_array(snps)


_POOL = None


def _row_blocks(n, parts=8):
    step = -(-n // parts)
    return [(a, min(a + step, n)) for a in range(0, n, step)]


def _pool():
    global _POOL
    if _POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=8)
    return _POOL


def scale_k(k, verbose=False):
    """kinship.py:94-100 -- c = tr(K) - sum(K)/n, K * (n-1)/c.  Host fp64, O(N^2)."""
    if isinstance(k, _lib.DeviceKinship):                  # in HBM: scaled where it lies (mmg_kin_acc_scale_k), once --
        if not k.scaled:                                   # the rule maps a scaled matrix onto itself (factor 1 +- 1e-16)
            k.acc.scale_k()
            k.scaled = True
        return k
    k = np.asarray(k, dtype=np.float64)
    n = len(k)
    if n <= 2048:
        c = np.sum((np.eye(n) - (1.0 / n) * np.ones(k.shape)) * k)      # as the reference writes it (:95)
        scalar = (n - 1) / c
        if verbose:
            print('Kinship scaled by: %0.4f' % scalar)
        return scalar * k
    # the same number, sum_ij (d_ij - 1/n) K_ij, without five N x N temporaries (1 GB and 0.1 s of the 0.44 s an
    # emmax() call took at N = 5000; 60 GB at N = 50,000); differs from the line above by summation order only.  Round 4:
    # the two passes over the matrix (sum, product) run in 8 fixed row blocks on a thread pool (numpy releases the GIL):
    # 22 -> 6 ms at N = 5000, a sixth of what is left of an emmax() call; fixed blocks = the same bits every run.
    blocks = _row_blocks(n)
    total = float(sum(_pool().map(lambda ab: float(np.sum(k[ab[0]:ab[1]])), blocks)))
    c = float(np.trace(k)) - total / n
    scalar = (n - 1) / c
    if verbose:
        print('Kinship scaled by: %0.4f' % scalar)
    out = np.empty_like(k)
    list(_pool().map(lambda ab: np.multiply(k[ab[0]:ab[1]], scalar, out=out[ab[0]:ab[1]]), blocks))
    return out


def calc_ibs_kinship(snps, snps_data_format='binary', snp_dtype='int8', dtype='single',
                     chunk_size=None, scaled=True, ctx=None, geno=None, keep_device=False, comm=None, m_total=None):
    """kinship.py:14-56 ('binary'): K = sum_m (2s-1)(2s-1)^T / (2M) + 0.5, then scale_k.

    The count matrix is an exact int8-MFMA GEMM on the device (bit-exact with the reference's
    float64 accumulator); chunk_size is accepted for signature compatibility (the device kernel
    tiles the SNP axis itself).  `geno` may pass an already-resident device genotype store.
    keep_device ('binary' only): return a _lib.DeviceKinship -- the matrix stays in HBM (mmg_kin_acc_set_ibs) and goes into
    LinearMixedModel.add_random_effect / emmax() without visiting the host; numpy sees it through __array__ / .host().
    comm / m_total: the SNP blocks of all ranks (RCCL sum of the counts)."""
    if snps_data_format not in ('binary', 'diploid_int'):
        raise NotImplementedError(snps_data_format)
    ctx = ctx or _lib.get_context()
    own = geno is None
    g = ctx.geno(_as_snp_matrix(snps)) if own else geno
    try:
        num_snps = g.M
        if keep_device and snps_data_format == 'binary' and hasattr(ctx, 'kinship_accumulator'):
            acc = ctx.kinship_accumulator(g.N)
            if hasattr(acc, 'set_ibs'):
                acc.set_ibs(g, scaled=scaled, comm=comm, m_total=m_total)
                return _lib.DeviceKinship(acc, scaled=scaled)
            acc.close()
        if snps_data_format == 'diploid_int':
            # kinship.py:33-41: k_ij = #(|a-b| = 0) + 0.5 #(|a-b| = 1) = M - 0.5 sum_m |a_m - b_m| for
            # 0/1/2 genotypes; |a-b| = a + b - 2 min(a,b) and min(a,b) = [a>=1][b>=1] + [a>=2][b>=2]:
            # two exact indicator GEMMs on the int8 matrix cores replace the O(N^2 M) bincount loop.
            if g.N > 2048 and hasattr(ctx, 'kinship_ibs_diploid'):
                return ctx.kinship_ibs_diploid(g, scaled=scaled)         # the same arithmetic in HBM (see the binary case below)
            c12 = ctx.kinship_indicator_counts(g, 1) + ctx.kinship_indicator_counts(g, 2)
            r = np.diag(c12).astype(np.float64)
            absdiff = r[:, None] + r[None, :] - 2.0 * c12
            k_mat = float(num_snps) - 0.5 * absdiff
            np.fill_diagonal(k_mat, 0.0)                   # the reference only fills i != j (:34-41)
            k_mat = k_mat / float(num_snps) + np.eye(g.N)  # :51
            return scale_k(k_mat) if scaled else k_mat
        if g.N > 2048 and hasattr(ctx, 'kinship_ibs'):
            # counts -> K -> scale_k in HBM, one download of doubles (mmg_kinship_ibs_f64): the host passes over N^2 values cost
            # more than the exact-count GEMM (50 of 92 ms at N = 5000 on 256 cores, seconds on a laptop's).  Below 2049 the
            # host keeps the reference's own expression of scale_k (bit for bit; 13 ms at N = 1000)
            return ctx.kinship_ibs(g, scaled=scaled)
        counts = ctx.kinship_ibs_counts(g)
    finally:
        if own:
            g.close()
    k_mat = counts.astype(np.float64) / (2 * float(num_snps)) + 0.5
    if scaled:
        k_mat = scale_k(k_mat)
    return k_mat


def calc_ibd_kinship(snps, dtype='single', scaled=True, ctx=None, geno=None):
    """kinship.py:59-75: z = (s - mean)/std per SNP (population std), K = sum z z^T / M, scale_k.

    Exact route (mmg_kin_acc_add_grm): z z' = a^2 s s' + a b (s 1' + 1 s') + b^2 1 1' with the per-SNP weight
    a^2 = 1/std^2 split into int8 digits -- 4-5 exact int8-MFMA GEMMs instead of the fp32-MFMA GEMM
    (`ctx.kinship_affine(g, 1/std, -mean/std)`, the north star's stated kernel, stays available and is what bench.py
    times).  A monomorphic SNP has std = 0: the reference divides by zero there and asserts (:67); here it raises."""
    ctx = ctx or _lib.get_context()
    own = geno is None
    g = ctx.geno(_as_snp_matrix(snps)) if own else geno
    try:
        acc = ctx.kinship_accumulator(g.N)
        try:
            acc.add_grm(g)
            if scaled and g.N > 2048 and hasattr(acc, 'scale_k'):
                # scale_k(K / n) = scale_k(K): scaled where the sum lies, one download (as hdf5_data._ibd_kinship does);
                # below 2049 individuals the host keeps the reference's own expression of scale_k
                acc.scale_k()
                return acc.fetch()[0]
            k_mat, num_snps = acc.fetch()
        finally:
            acc.close()
    finally:
        if own:
            g.close()
    k_mat = k_mat / float(num_snps)
    if scaled:
        k_mat = scale_k(k_mat)
    return k_mat


def prepare_k(k, k_accessions, accessions):
    """kinship.py:79-90."""
    if k_accessions == accessions:
        return np.asarray(k)
    indices_to_keep = []
    for acc in accessions:
        try:
            indices_to_keep.append(k_accessions.index(acc))
        except ValueError:
            continue
    k = np.asarray(k)
    return k[indices_to_keep, :][:, indices_to_keep]


def load_kinship_from_file(kinship_file, accessions=None, scaled=True):
    """kinship.py:145-158 -- datasets 'kinship', 'accessions', 'n_snps' of a kinship container (a chunkstore
    directory, or an HDF5 file when h5py is installed)."""
    import os
    from . import chunkstore
    assert os.path.exists(kinship_file), 'File not found.'
    f = chunkstore.open_container(kinship_file, 'r')
    k = np.asarray(f['kinship'][...], dtype=np.float64)
    k_accessions = [a.decode() if isinstance(a, bytes) else str(a) for a in np.asarray(f['accessions'][...]).reshape(-1)]
    n_snps = int(np.asarray(f['n_snps'][...]))
    f.close()
    if accessions:
        k = prepare_k(k, k_accessions, [str(a) for a in accessions])
    if scaled:
        k = scale_k(k)
    return {'k': k, 'accessions': k_accessions, 'n_snps': n_snps}


def save_kinship_to_file(kinship_file, kinship_mat, k_accessions, n_snps):
    """kinship.py:162-167."""
    from . import chunkstore
    f = chunkstore.open_container(kinship_file, 'w')
    f.create_dataset('kinship', data=np.asarray(kinship_mat))
    f.create_dataset('accessions', data=np.asarray([str(a) for a in k_accessions], dtype='S'))
    f.create_dataset('n_snps', data=np.array(n_snps))
    f.close()


def save_kinship_in_text_format(filename, k, accessions):
    """kinship.py:170-173."""
    with open(filename, 'w') as f:
        for acc, row in zip(accessions, np.asarray(k)):
            f.write('%s,%s\n' % (acc, ','.join(map(str, row.tolist()))))

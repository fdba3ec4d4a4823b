"""CPU oracle for the mixmogam EMMAX hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The shipped path (mixmogam_amd/) never imports it and fails loudly when the HIP library is
missing.

What it is: a float64 numpy/scipy restatement of the reference's algorithm for
    kinship.calc_ibs_kinship / calc_ibd_kinship / scale_k        (/root/reference/kinship.py)
    LinearMixedModel._get_eigen_L_/_get_eigen_R_/get_estimates   (/root/reference/linear_models.py)
    LinearMixedModel._emmax_f_test_ / _emmax_permutations_       (/root/reference/linear_models.py)
each function citing the reference file:line it follows.  Third-party arithmetic the reference
delegates to (LAPACK via scipy.linalg.eigh/lstsq/qr/pinv, scipy.optimize.newton,
scipy.stats.f.sf; no versions pinned by the reference, README:9-12) is taken from the scipy in
this image (1.15.3) exactly as the reference calls it.

Parity pinning: the reference ships no tests or golden vectors, so this
oracle is pinned against OUTPUTS OF THE REFERENCE ITSELF, run in the build container through
tests/golden/refshim.py (2to3 on the fly, nothing copied) and committed as tests/golden/*.npz
by tests/golden/make_golden.py.  tests/test_oracle_golden.py checks every function here against
those vectors (double-promoted reference: <=1e-9 relative; literal fp32 reference: loose,
measured tolerance).
"""
import warnings

import numpy as np
from scipy import linalg, optimize, stats


# ----------------------------------------------------------------------------- kinship
def scale_k(k):
    """kinship.py:94-100: c = sum((I - 11'/n) * K) elementwise = tr(K) - sum(K)/n; K*(n-1)/c."""
    k = np.asarray(k, dtype=np.float64)
    n = len(k)
    c = np.sum((np.eye(n) - (1.0 / n) * np.ones(k.shape)) * k)
    return ((n - 1) / c) * k


def ibs_counts(snps, chunk_size=None):
    """kinship.py:29-44 ('binary'): C = sum over chunks of (2S-1)(2S-1)^T.  Exact integers.

    snps: [M x N] int8, SNP-major.  Returns int64 [N x N]."""
    snps = np.asarray(snps)
    m, n = snps.shape
    if chunk_size is None:
        chunk_size = n
    c = np.zeros((n, n), dtype=np.float64)
    for i in range(0, m, chunk_size):
        sm = snps[i:i + chunk_size].T.astype(np.float64) * 2.0 - 1.0
        c = c + sm @ sm.T
    return np.rint(c).astype(np.int64)


def calc_ibs_kinship(snps, scaled=True, chunk_size=None):
    """kinship.py:14-56 binary branch: K = C/(2M) + 0.5 (:53), then scale_k (:55). float64
    (the reference's accumulator silently becomes float64)."""
    snps = np.asarray(snps)
    m = snps.shape[0]
    k = ibs_counts(snps, chunk_size).astype(np.float64) / (2 * float(m)) + 0.5
    return scale_k(k) if scaled else k


def calc_ibd_kinship(snps, scaled=True, acc_dtype=np.float64):
    """kinship.py:59-75: per-SNP standardise (population std, :66), K += X^T X per chunk of N
    SNPs (:69), /M (:72), scale_k.  acc_dtype=float32 reproduces the literal 'single'
    accumulator (:62)."""
    snps = np.asarray(snps)
    m, n = snps.shape
    k = np.zeros((n, n), dtype=acc_dtype)
    for i in range(0, m, n):
        a = snps[i:i + n].T.astype(np.float64)            # N x chunk
        z = (a - a.mean(0)) / a.std(0)
        k += (z @ z.T).astype(acc_dtype)
    k = k / float(m)
    return scale_k(k) if scaled else np.asarray(k, dtype=np.float64)


# ----------------------------------------------------------------------------- eigen + REML
def eig_L(K):
    """linear_models.py:589-596: eigh(K), ascending; vectors returned TRANSPOSED (rows)."""
    evals, evecs = linalg.eigh(np.asarray(K, dtype=np.float64))
    return {'values': evals, 'vectors': evecs.T.copy()}


def eig_R(X, K):
    """linear_models.py:600-615: S = I - X pinv(X'X) X'; eigh(S (K+I) S); drop first q; -1."""
    X = np.asarray(X, dtype=np.float64)
    K = np.asarray(K, dtype=np.float64)
    n, q = X.shape
    hat = X @ linalg.pinv(X.T @ X) @ X.T
    S = np.eye(n) - hat
    Mm = S @ (K + np.eye(n)) @ S
    evals, evecs = linalg.eigh(Mm)
    return {'values': evals[q:] - 1.0, 'vectors': evecs.T[q:].copy()}


def _rell(delta, eig_vals, sq_etas):
    """linear_models.py:618-623."""
    p = len(eig_vals)
    c_1 = 0.5 * p * (np.log(p / (2.0 * np.pi)) - 1)
    v = eig_vals + delta
    return c_1 - 0.5 * (p * np.log(np.sum(sq_etas / v)) + np.sum(np.log(v)))


def _redll(delta, eig_vals, sq_etas):
    """linear_models.py:626-631."""
    p = len(eig_vals)
    v1 = eig_vals + delta
    v2 = sq_etas / v1
    return p * np.sum(v2 / v1) / np.sum(v2) - np.sum(1.0 / v1)


def _ll(delta, eig_vals, eig_vals_L, sq_etas):
    """linear_models.py:634-640."""
    n = len(eig_vals_L)
    c_1 = 0.5 * n * (np.log(n / (2.0 * np.pi)) - 1)
    return c_1 - 0.5 * (n * np.log(np.sum(sq_etas / (eig_vals + delta))) + np.sum(np.log(eig_vals_L + delta)))


def _dll(delta, eig_vals, eig_vals_L, sq_etas):
    """linear_models.py:643-649."""
    n = len(eig_vals_L)
    v1 = eig_vals + delta
    v2 = sq_etas / v1
    return n * np.sum(v2 / v1) / np.sum(v2) - np.sum(1.0 / (eig_vals_L + delta))


def get_estimates(y, X, K, eigL=None, eigR=None, ngrids=50, llim=-10, ulim=10, esp=1e-6, method='REML'):
    """linear_models.py:771-912, xs=None; method 'REML' (:803-810) or 'ML' (:811-825).

    Returns dict(max_ll, delta, beta, ve, vg, rss, mahalanobis_rss, H_sqrt_inv,
    pseudo_heritability)."""
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    X = np.asarray(X, dtype=np.float64)
    n, q = X.shape
    if eigL is None:
        eigL = eig_L(K)
    if eigR is None:
        eigR = eig_R(X, K)                                   # :787-788
    p = n - q
    m = ngrids + 1
    etas = eigR['vectors'] @ y                               # :794
    sq_etas = etas * etas
    log_deltas = (np.arange(m, dtype=np.float64) / ngrids) * (ulim - llim) + llim   # :796
    deltas = np.exp(log_deltas)
    eig_vals = eigR['values']
    lambdas = eig_vals[:, None] + deltas[None, :]            # :802  (p x m)
    s1 = np.sum(sq_etas[:, None] / lambdas, axis=0)
    s2 = np.sum(np.log(lambdas), axis=0)
    lls = 0.5 * (p * (np.log(p / (2.0 * np.pi)) - 1 - np.log(s1)) - s2)           # :807
    s3 = np.sum(sq_etas[:, None] / (lambdas * lambdas), axis=0)
    s4 = np.sum(1 / lambdas, axis=0)
    dlls = 0.5 * (p * s3 / s1 - s4)                                                # :810
    if method == 'ML':                                                             # :811-825
        xis = eigL['values'][:, None] + deltas[None, :]
        s2 = np.sum(np.log(xis), axis=0)
        lls = 0.5 * (n * (np.log(n / (2.0 * np.pi)) - 1 - np.log(s1)) - s2)
        s4 = np.sum(1 / xis, axis=0)
        dlls = 0.5 * (n * s3 / s1 - s4)
    max_ll_i = int(np.argmax(lls))
    max_ll = lls[max_ll_i]
    zero_intervals = []
    last_dll, last_ll = dlls[0], lls[0]
    for i in range(1, len(dlls)):                            # :832-836
        if dlls[i] < 0 and last_dll > 0:
            zero_intervals.append(((lls[i] + last_ll) * 0.5, i))
        last_ll, last_dll = lls[i], dlls[i]
    if zero_intervals:
        opt_ll, opt_i = max(zero_intervals)
        opt_delta = 0.5 * (deltas[opt_i - 1] + deltas[opt_i])
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                if method == 'REML':
                    new_opt_delta = optimize.newton(_redll, opt_delta, args=(eig_vals, sq_etas),
                                                    tol=esp, maxiter=100)          # :847
                else:
                    new_opt_delta = optimize.newton(_dll, opt_delta, args=(eig_vals, eigL['values'], sq_etas),
                                                    tol=esp, maxiter=100)          # :849
        except Exception:
            new_opt_delta = opt_delta
        if opt_i > 1 and deltas[opt_i - 1] - esp < new_opt_delta < deltas[opt_i] + esp:
            opt_delta = new_opt_delta
        elif opt_i == 1 and 0.0 < new_opt_delta < deltas[opt_i] + esp:
            opt_delta = new_opt_delta
        elif opt_i == len(deltas) - 1 and new_opt_delta > deltas[opt_i - 1] - esp \
                and not np.isinf(new_opt_delta):
            opt_delta = new_opt_delta
        opt_ll = _rell(opt_delta, eig_vals, sq_etas) if method == 'REML' \
            else _ll(opt_delta, eig_vals, eigL['values'], sq_etas)                 # :880-884
        if opt_ll < max_ll:                                  # :886
            opt_delta = deltas[max_ll_i]
    else:
        opt_delta = deltas[max_ll_i]
        opt_ll = max_ll
    # :894-895 -- the reference divides a (p,1) column by a (p,) vector, which broadcasts to a
    # p x p outer quotient ("BUG NEEDS TO BE FIXED HERE!!!" in its own comment); vg as RETURNED
    # by the reference is therefore sum(sq_etas) * sum(1/(lambda+delta)) / p.  Restated as is.
    l = sq_etas[:, None] / (eig_vals + opt_delta)[None, :]
    opt_vg = np.sum(l) / p                                   # :895
    opt_ve = opt_vg * opt_delta
    H_sqrt_inv = (1.0 / np.sqrt(eigL['values'] + opt_delta))[:, None] * eigL['vectors']   # :898
    X_t = H_sqrt_inv @ X
    Y_t = H_sqrt_inv @ y
    beta_est, mahal, _, _ = linalg.lstsq(X_t, Y_t)
    resid = y - X @ beta_est
    return {'max_ll': opt_ll, 'delta': opt_delta, 'beta': beta_est, 've': opt_ve, 'vg': opt_vg,
            'rss': float(resid @ resid), 'mahalanobis_rss': float(np.sum((Y_t - X_t @ beta_est) ** 2)),
            'H_sqrt_inv': H_sqrt_inv, 'pseudo_heritability': 1.0 / (1 + opt_delta)}


# ----------------------------------------------------------------------------- p-values
def f_sf(f_stats, dfn, dfd):
    """scipy.stats.f.sf as called at linear_models.py:1349,1172."""
    return stats.f.sf(f_stats, dfn, dfd)


# ----------------------------------------------------------------------------- EMMAX scan
def scan_prepare(y, X, H_sqrt_inv, Z=None):
    """The SNP-independent part of _emmax_f_test_ (linear_models.py:1290-1303).

    Returns dict with h0_rss, h0_betas, r (residualised transformed phenotype), Mp
    (= H'(I - QQ'), :1303), w = Mp r and A = Mp Mp' (the closed form of SURVEY 8a-a9)."""
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    X = np.asarray(X, dtype=np.float64)
    H = np.asarray(H_sqrt_inv, dtype=np.float64)
    h0_X = H @ X                                             # :1290
    Y = H @ y                                                # :1291
    h0_betas, _, _, _ = linalg.lstsq(h0_X, Y)                # :1292
    r = Y - h0_X @ h0_betas                                  # :1293
    h0_rss = float(r @ r)
    if Z is not None:
        H = H @ np.asarray(Z, dtype=np.float64)              # :1296-1297
    Q, _ = linalg.qr(h0_X, mode='economic')                  # :1300
    Mp = H.T @ (np.eye(len(y)) - Q @ Q.T)                    # :1303
    return {'h0_rss': h0_rss, 'h0_betas': h0_betas, 'r': r, 'Mp': Mp,
            'w': Mp @ r, 'A': Mp @ Mp.T, 'n': len(y), 'q': X.shape[1]}


def scan_closed(snps, prep):
    """Closed form of the per-SNP loop linear_models.py:1316-1349:
       num = (s.w)^2, den = s'As, rss = h0_rss - num/den, F = (h0_rss/rss - 1)(n-q-1),
       p = f.sf(F, 1, n-q-1).  A SNP with den == 0 (monomorphic after projection) keeps
       rss = h0_rss (-> F = 0, p = 1), as :1308,:1329 leave it."""
    S = np.asarray(snps, dtype=np.float64)
    num = (S @ prep['w']) ** 2
    den = np.einsum('ij,ij->i', S @ prep['A'], S)
    h0 = prep['h0_rss']
    ok = den > 1e-12 * np.abs(den).max() if len(den) else den > 0
    rss = np.where(ok, h0 - num / np.where(ok, den, 1.0), h0)
    n_p = prep['n'] - prep['q'] - 1
    rss_ratio = h0 / rss
    f = (rss_ratio - 1) * n_p
    return {'rss': rss, 'f_stats': f, 'ps': f_sf(f, 1, n_p), 'var_perc': 1 - 1 / rss_ratio,
            'h0_rss': h0, 'h0_betas': prep['h0_betas'], 'num': num, 'den': den}


def scan_loop(snps, prep, dtype=np.float64):
    """Literal loop structure of linear_models.py:1315-1349: chunks of N SNPs, `chunk @ Mp`
    GEMM in `dtype`, one scipy.linalg.lstsq per SNP, vectorised f.sf.  dtype=float32 is the
    reference's 'single' path; this function is also the timed CPU baseline (bench.py)."""
    n = prep['n']
    Mp = prep['Mp'].astype(dtype)
    Y = prep['r'].astype(dtype).reshape(-1, 1)
    h0 = prep['h0_rss']
    snps = np.asarray(snps)
    m = len(snps)
    rss_list = np.repeat(h0, m).astype(np.float64)
    for i in range(0, m, n):
        Xs = snps[i:i + n].astype(dtype) @ Mp
        for j in range(len(Xs)):
            _, rss, _, _ = linalg.lstsq(Xs[j].reshape(-1, 1), Y, overwrite_a=True)
            if np.size(rss) and rss[0]:
                rss_list[i + j] = rss[0]
    n_p = n - prep['q'] - 1
    rss_ratio = h0 / rss_list
    f = (rss_ratio - 1) * n_p
    return {'rss': rss_list, 'f_stats': f, 'ps': f_sf(f, 1, n_p), 'var_perc': 1 - 1 / rss_ratio,
            'h0_rss': h0, 'h0_betas': prep['h0_betas']}


def emmax(snps, y, K, cofactors=None, Z=None):
    """linear_models.py:1790-1816 + :1233-1267 (emma_num=0, with_betas=False):
    scale_k again (:580), eig_L, eig_R, REML, scan.  Z: incidence matrix of replicated measurements
    [n_values x n_individuals] -- the random effect becomes Z K Z' (:1796), cofactors Z c (:1799), and the
    SNPs are expanded through H Z (:1296-1297)."""
    y = np.asarray(y, dtype=np.float64)
    n = len(y)
    X = np.ones((n, 1))
    if Z is not None:
        Z = np.asarray(Z, dtype=np.float64)
        K = Z @ np.asarray(K, dtype=np.float64) @ Z.T
    if cofactors is not None:
        for c in cofactors:
            c = np.asarray(c, dtype=np.float64).reshape(-1)
            X = np.hstack([X, (Z @ c if Z is not None else c).reshape(n, 1)])
    Ks = scale_k(K)
    est = get_estimates(y, X, Ks)
    prep = scan_prepare(y, X, est['H_sqrt_inv'], Z=Z)
    res = scan_closed(snps, prep)
    for k in ('pseudo_heritability', 've', 'vg', 'max_ll', 'delta'):
        res[k] = est[k]
    return res


def emmax_multi(snps, ys, K, cofactors=None):
    """Several phenotypes over the same genotypes and kinship = the reference's loop of emmax() calls
    (one LinearMixedModel, REML and scan per phenotype: phenotypeData.py:70-78, hdf5_data.py:262-330)."""
    res = [emmax(snps, y, K, cofactors=cofactors) for y in np.asarray(ys, dtype=np.float64)]
    out = {k: np.asarray([r[k] for r in res]) for k in ('ps', 'f_stats', 'rss', 'var_perc')}
    for k in ('h0_rss', 'pseudo_heritability', 'max_ll', 'delta'):
        out[k] = np.asarray([float(np.asarray(r[k]).reshape(-1)[0]) for r in res])
    return out


def linear_model(snps, y, cofactors=None):
    """linear_models.py:3168-3183 -> LinearModel.fast_f_test (:196-257): the scan with H = I."""
    y = np.asarray(y, dtype=np.float64)
    n = len(y)
    X = np.ones((n, 1))
    if cofactors is not None:
        for c in cofactors:
            X = np.hstack([X, np.asarray(c, dtype=np.float64).reshape(n, 1)])
    prep = scan_prepare(y, X, np.eye(n))
    return scan_closed(snps, prep)


def ibs_diploid_unscaled(snps):
    """kinship.py:33-41,51 ('diploid_int', scaled=False): k_ij = #(|a-b|=0) + 0.5 #(|a-b|=1) for
    i != j, divided by M, plus the identity."""
    S = np.asarray(snps, dtype=np.int64)
    m, n = S.shape
    k = np.zeros((n, n))
    for i in range(n):
        d = np.abs(S[:, i:i + 1] - S)                       # M x N
        k[i] = (d == 0).sum(0) + 0.5 * (d == 1).sum(0)
    np.fill_diagonal(k, 0.0)
    return k / float(m) + np.eye(n)


def exact_emma(snps, y, X, K, eigL=None, ngrids=50, llim=-4, ulim=10, esp=1e-6):
    """expedited_REML_t_test (linear_models.py:931-968): get_estimates with xs = snp per SNP
    (:782-788, :914-926).  K must already be scaled as the model holds it."""
    y = np.asarray(y, dtype=np.float64)
    X = np.asarray(X, dtype=np.float64)
    n = len(y)
    if eigL is None:
        eigL = eig_L(K)
    out = {k: [] for k in ('ps', 'f_stats', 'rss', 'var_perc', 'vgs', 'ves', 'max_lls', 'betas')}
    for s in np.asarray(snps, dtype=np.float64):
        Xf = np.hstack([X, s.reshape(n, 1)])
        est = get_estimates(y, Xf, K, eigL=eigL, eigR=None, ngrids=ngrids, llim=llim, ulim=ulim, esp=esp)
        H = est['H_sqrt_inv']
        h0_X = H @ X
        Yt = H @ y
        b0, _, _, _ = linalg.lstsq(h0_X, Yt)
        h0_rss = float(np.sum((Yt - h0_X @ b0) ** 2))
        p = n - Xf.shape[1]
        f = (h0_rss / est['mahalanobis_rss'] - 1) * p            # :921 (one SNP column)
        out['f_stats'].append(f)
        out['ps'].append(float(stats.f.sf(f, 1, p)))
        out['rss'].append(est['rss'])
        out['var_perc'].append(1.0 - est['mahalanobis_rss'] / h0_rss)
        out['vgs'].append(est['vg'])                             # :955-963
        out['ves'].append(est['ve'])
        out['max_lls'].append(est['max_ll'])
        out['betas'].append(np.asarray(est['beta']).reshape(-1))
    return {k: np.asarray(v) for k, v in out.items()}


def emmax_with_emma(snps, y, K, cofactors=None, emma_num=15):
    """emmax(..., emma_num > 0): EMMAX scan, then the emma_num smallest p-values are replaced by the
    exact-EMMA values (linear_models.py:1365-1377)."""
    res = emmax(snps, y, K, cofactors=cofactors)
    y = np.asarray(y, dtype=np.float64)
    n = len(y)
    X = np.ones((n, 1))
    if cofactors is not None:
        for c in cofactors:
            X = np.hstack([X, np.asarray(c, dtype=np.float64).reshape(n, 1)])
    order = np.argsort(res['ps'], kind='stable')[:emma_num]
    top = exact_emma(np.asarray(snps)[order], y, X, scale_k(K))
    for k in ('ps', 'f_stats', 'rss', 'var_perc'):
        res[k] = np.array(res[k], dtype=np.float64)
        res[k][order] = top[k]
    return res


# ----------------------------------------------------------------------------- permutations
def perm_prepare(y, X, H_sqrt_inv, perm_idx):
    """SNP-independent part of _emmax_permutations_ (linear_models.py:1135-1156).
    perm_idx: [P x N] int index matrix; column p of Ys is r[perm_idx[p]] (the reference
    shuffles in place with the global RNG, :1151-1154 -- callers record/replay the indices)."""
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    X = np.asarray(X, dtype=np.float64)
    H = np.asarray(H_sqrt_inv, dtype=np.float64)
    y = y - y.mean()                                         # :1140
    h0_X = H @ X
    Y = H @ y
    h0_betas, _, _, _ = linalg.lstsq(h0_X, Y)                # :1143
    r = Y - h0_X @ h0_betas                                  # :1144
    h0_rss = float(r @ r)
    r = r - h0_X @ h0_betas                                  # :1147 (second subtraction, kept)
    Ys = np.stack([r[np.asarray(ix)] for ix in perm_idx], axis=1)   # N x P
    return {'h0_rss': h0_rss, 'Ys': Ys, 'H': H, 'n': len(y), 'q': X.shape[1],
            'yy': np.einsum('ij,ij->j', Ys, Ys)}


def perm_closed(snps, pp):
    """linear_models.py:1157-1175: row-centre SNPs (:1159), t = H s~ (:1160; no covariate
    projection), rss_{m,p} = Ys_p.Ys_p - (t.Ys_p)^2/(t.t) (:1163), running min over SNPs
    (:1164) starting from h0_rss (:1156), max F / min p (:1171-1172)."""
    S = np.asarray(snps, dtype=np.float64)
    S = S - S.mean(axis=1, keepdims=True)
    T = S @ pp['H'].T                                        # M x N
    tt = np.einsum('ij,ij->i', T, T)
    G = T @ pp['Ys']                                         # M x P
    ok = tt > 0
    rss = pp['yy'][None, :] - np.where(ok[:, None], G * G / np.where(ok, tt, 1.0)[:, None], 0.0)
    min_rss = np.minimum(pp['h0_rss'], rss.min(axis=0)) if len(S) else np.repeat(pp['h0_rss'], G.shape[1])
    n_p = pp['n'] - pp['q'] - 1
    max_f = (pp['h0_rss'] / min_rss - 1.0) * n_p
    return {'min_rss': min_rss, 'max_f_stats': max_f, 'min_ps': f_sf(max_f, 1, n_p)}


def transformed_snps(snps, X, H_sqrt_inv):
    """t_snps of _emmax_f_test_(return_transformed_snps=True), linear_models.py:1300-1303,1316-1321:
    M = H'(I - QQ') with Q from qr(H X); row m is s_m M."""
    H = np.asarray(H_sqrt_inv, dtype=np.float64)
    Q, _ = linalg.qr(H @ np.asarray(X, dtype=np.float64), mode='economic')
    Mp = H.T @ (np.eye(len(H)) - Q @ Q.T)
    return np.asarray(snps, dtype=np.float64) @ Mp


def perm_public(snps, y, X, H_sqrt_inv, perm_idx, reference_indexing=True):
    """LinearMixedModel.emmax_permutations, linear_models.py:1180-1230 (the worker of emmax_perm_test, :1819):
    Y = H y not centred, ONE subtraction of the null fit (:1198), Xs = S H' then Xs - mean(Xs) per SNP (:1209-1211),
    rss_{m,p} by lstsq(Xs[m], Ys) (:1212).  reference_indexing: the reference's literal result -- it writes
    rss_list.min() (min over the permutations of SNP m) into slot m of a per-permutation array (:1213), which needs
    num_snps <= num_perm; False: min over the SNPs per permutation, what the docstring describes."""
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    X = np.asarray(X, dtype=np.float64)
    H = np.asarray(H_sqrt_inv, dtype=np.float64)
    n = len(y)
    n_p = n - (X.shape[1] + 1)
    Y = H @ y
    h0_X = H @ X
    h0_betas, _, _, _ = linalg.lstsq(h0_X, Y)
    r = Y - h0_X @ h0_betas
    h0_rss = float(r @ r)
    Ys = np.stack([r[np.asarray(ix)] for ix in perm_idx], axis=1)            # n x P
    P = Ys.shape[1]
    T = np.asarray(snps, dtype=np.float64) @ H.T
    T = T - T.mean(axis=1, keepdims=True)
    tt = np.einsum('ij,ij->i', T, T)
    G = T @ Ys
    ok = tt > 0
    rss = np.einsum('ij,ij->j', Ys, Ys)[None, :] - np.where(ok[:, None], G * G / np.where(ok, tt, 1.0)[:, None], 0.0)
    if reference_indexing:
        if len(T) > P:
            raise IndexError("linear_models.py:1213 indexes a per-permutation array by SNP")
        min_rss = np.repeat(h0_rss, P).astype(np.float64)
        min_rss[:len(T)] = rss.min(axis=1)
    else:
        min_rss = np.minimum(h0_rss, rss.min(axis=0)) if len(T) else np.repeat(h0_rss, P)
    max_f = (h0_rss / min_rss - 1.0) * n_p
    return {'min_rss': min_rss, 'max_f_stats': max_f, 'min_ps': f_sf(max_f, 1, n_p), 'h0_rss': h0_rss}


# ----------------------------------------------------------------------------- HDF5 drivers (hdf5_data.py)
def hdf5_maf_filter(freqs, min_maf):
    """hdf5_data.py:91-93 (= :153-155, :213-215, :249-251, :301-303): keep where min(f, 1 - f) > min_maf, strictly."""
    freqs = np.asarray(freqs, dtype=np.float64)
    return np.minimum(freqs, 1 - freqs) > min_maf


def hdf5_ibd_kinship(chroms, chunk_size=1000, min_maf=None, acc_dtype=np.float64, normalised=None):
    """The kinship loop the three drivers share -- calculate_ibd_kinship hdf5_data.py:30-58 (no filter) and
    run_emmax :84-111 / run_emmax_perm :205-232 (MAF-filtered rows): per chromosome in key order, chunks of
    chunk_size rows WITHIN the chromosome (:99), each SNP standardised with the population std (:103),
    k_mat += x'x in the accumulator's dtype ('single' as written, :84), / n_snps (:107), then scale_k's rule inline
    (:108-111).  chroms: ordered [(raw_snps [M_c x N], freqs [M_c]), ...].  normalised: per chromosome None or the file's
    pre-normalised `snps` dataset, which calculate_ibd_kinship takes as it is (:36-38; the loop bound stays len(raw_snps)).
    Returns (k, n_snps)."""
    n = np.asarray(chroms[0][0]).shape[1]
    k_mat = np.zeros((n, n), dtype=acc_dtype)
    n_snps = 0
    for ci, (snps, freqs) in enumerate(chroms):
        snps = np.asarray(snps)
        pre = None if normalised is None else normalised[ci]
        if min_maf is not None:
            snps = snps[hdf5_maf_filter(freqs, min_maf)]
        for i in range(0, len(snps), chunk_size):
            if pre is not None:
                x = np.asarray(pre[i:i + chunk_size], dtype=np.float64)          # :38
            else:
                x = snps[i:i + chunk_size].T.astype(np.float64)
                x = ((x - x.mean(0)) / x.std(0)).T
            n_snps += len(x)
            k_mat += (x.T @ x).astype(acc_dtype)
    k_mat = k_mat / float(n_snps)
    c = np.sum((np.eye(n) - (1.0 / n) * np.ones(k_mat.shape)) * np.asarray(k_mat, dtype=np.float64))
    return ((n - 1) / c) * k_mat, n_snps


def hdf5_run_emmax(chroms, positions, y, min_maf=0.1, chunk_size=1000, k=None, perm_idx=None, perm_H=None):
    """run_emmax hdf5_data.py:70-187 and, with perm_idx [P x N], run_emmax_perm :191-351.
    Kinship from the MAF-filtered rows unless k is given (:113-115, recalculate_kinship=False reads the stored one);
    LinearMixedModel(phenotypes) + add_random_effect(k) -- which scales k once more (linear_models.py:580);
    eig_L, eig_R, REML once (:126-137); _emmax_f_test_(emma_num=0) per chromosome on its filtered rows (:157-176).
    Permutation variant: the test runs on every chromosome but the LAST (`chr12_snps`, :294-311,330); both result
    arrays are stored sorted ascending (:339-341, the `[::-1]` is a no-op expression), the 5 % entries are index
    num_perm // 20 of each (:342-347); the stored `num_snps` is the LAST chromosome's filtered count (the counting loop
    reuses the name, :253,289), where run_emmax copies the input file's unfiltered total (:150).
    perm_H: the H_sqrt_inv to shuffle in (default: this model's own) -- a permutation's outcome depends on the signs of its
    rows (the ROTATED residuals are shuffled, linear_models.py:1151-1154), which are LAPACK's choice, so replaying a recorded
    reference run needs the reference's matrix."""
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    n = len(y)
    keep = [hdf5_maf_filter(f, min_maf) for _s, f in chroms]
    if k is None:
        k, _n = hdf5_ibd_kinship(chroms, chunk_size, min_maf)
    X = np.ones((n, 1))
    est = get_estimates(y, X, scale_k(np.asarray(k, dtype=np.float64)))
    prep = scan_prepare(y, X, est['H_sqrt_inv'])
    out = {'kinship': np.asarray(k), 'chrom_ps': [], 'chrom_positions': []}
    for k_ in ('pseudo_heritability', 've', 'vg', 'max_ll', 'delta'):
        out[k_] = est[k_]
    for (snps, _f), kp, pos in zip(chroms, keep, positions):
        out['chrom_ps'].append(scan_closed(np.asarray(snps)[kp], prep)['ps'])
        out['chrom_positions'].append(np.asarray(pos)[kp])
    out['num_snps'] = sum(len(np.asarray(s)) for s, _f in chroms)             # :150 (the input file's own dataset)
    if perm_idx is not None:
        chr12 = np.vstack([np.asarray(s)[kp] for (s, _f), kp in zip(chroms, keep)][:-1]).astype(np.float64)
        out['H_sqrt_inv'] = est['H_sqrt_inv']
        pr = perm_closed(chr12, perm_prepare(y, X, est['H_sqrt_inv'] if perm_H is None else perm_H, perm_idx))
        five = len(perm_idx) // 20
        out['perm_min_ps'] = np.sort(pr['min_ps'])
        out['perm_max_f_stats'] = np.sort(pr['max_f_stats'])
        out['five_perc_perm_min_ps'] = out['perm_min_ps'][five]
        out['five_perc_perm_max_f_stats'] = out['perm_max_f_stats'][five]
        out['num_snps'] = int(keep[-1].sum())                                 # :253,289
    return out


# ----------------------------------------------------------------------------- synthetic data
def hash_genotypes(m0, m1, n, seed, maf_q16=None):
    """Counter-based Bernoulli genotypes shared by bench.py (device generator
    mmg_geno_fill_hash) and the CPU baseline sample: bit = splitmix-style hash of
    (seed, snp, indiv) compared with a per-SNP threshold.  Restates simulations.py:21-23
    (round(U(0,1)) -> int8, SNP-major [M x N]) with an explicit, order-independent RNG."""
    snp = np.arange(m0, m1, dtype=np.uint64)[:, None]
    ind = np.arange(n, dtype=np.uint64)[None, :]
    with np.errstate(over='ignore'):
        x = (snp * np.uint64(0x9E3779B97F4A7C15)) ^ (ind * np.uint64(0xBF58476D1CE4E5B9)) \
            ^ (np.uint64(seed) * np.uint64(0x94D049BB133111EB))
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    u16 = (x >> np.uint64(48)).astype(np.uint32)
    thr = 32768 if maf_q16 is None else np.asarray(maf_q16, dtype=np.uint32)[:, None]
    return (u16 < thr).astype(np.int8)


def _hash3(seed, snp, ind):
    with np.errstate(over='ignore'):
        x = (snp * np.uint64(0x9E3779B97F4A7C15)) ^ (ind * np.uint64(0xBF58476D1CE4E5B9)) \
            ^ (np.uint64(seed) * np.uint64(0x94D049BB133111EB))
        x ^= x >> np.uint64(30)
        x *= np.uint64(0xBF58476D1CE4E5B9)
        x ^= x >> np.uint64(27)
        x *= np.uint64(0x94D049BB133111EB)
        x ^= x >> np.uint64(31)
    return x


def hash_genotypes_structured(m0, m1, n, seed, npop=3, spread_q16=9830):
    """Host twin of mmg_geno_fill_structured (k_pack.hip:fill_struct_kernel), bit for bit: `npop` contiguous
    populations, per-SNP ancestral frequency U[0.1, 0.9] + per-population deviation, Bernoulli genotypes."""
    snp = np.arange(m0, m1, dtype=np.uint64)
    s2 = np.uint64(seed) ^ np.uint64(0x5bf03635ca3d9a1f)
    anc = 6554 + (((_hash3(s2, snp, np.uint64(1000003)) >> np.uint64(48)) * np.uint64(52428)) >> np.uint64(16)).astype(np.int64)
    thr = np.empty((len(snp), npop), dtype=np.int64)
    for k in range(npop):
        z = np.full(len(snp), -2 * 65535, dtype=np.int64)
        for j in range(4):
            z += (_hash3(s2, snp, np.uint64(2000003 + 4 * k + j)) >> np.uint64(48)).astype(np.int64)
        thr[:, k] = np.clip(anc + ((np.int64(spread_q16) * z) >> np.int64(16)), 1311, 64225)
    pop = (np.arange(n, dtype=np.int64) * npop) // n
    u16 = (_hash3(seed, snp[:, None], np.arange(n, dtype=np.uint64)[None, :]) >> np.uint64(48)).astype(np.int64)
    return (u16 < thr[:, pop]).astype(np.int8)

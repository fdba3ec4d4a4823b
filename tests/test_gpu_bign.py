"""GPU test (-m gpu) of the route BASELINE config 5 takes in production: N beyond rocSOLVER's syevd index range
(N^2 >= 2^31, N > 46,340), where every N x N object is addressed with 64-bit indices, the REML sums come from one band
reduction of K (csrc/reml_band.hip: own panel-QR / fp64-MFMA kernels + `_64` rocBLAS GEMMs, banded factorisations per
delta) and the scan model from a Cholesky factorisation of K + delta I (csrc/reml_chol.hip: potrf / trmm / syrk `_64`)
instead of an eigendecomposition (linear_models.py:589-615,771-927 at hdf5_data.py:70-187's size).  Small-N tests with forced
thresholds cannot see a 32-bit index overflow; this one runs at N = 47,104 (N^2 = 2.22e9, 17.7 GB per fp64 matrix).

Reference arithmetic at this size: float64 conjugate-gradient solves with H = K + delta I on the host (H is well
conditioned for the delta values used), i.e. no factorisation and no code shared with the device route."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

orc = pytest.importorskip("oracle.emmax_oracle")

N, M = 47104, 50000


@pytest.fixture(scope="module")
def ctx():
    from mixmogam_amd import _lib
    return _lib.get_context()


def _cg(K, delta, B, tol=1e-13, maxit=300):
    """(K + delta I) X = B, one conjugate-gradient recurrence per column, all columns through the same matrix product."""
    X = np.zeros_like(B)
    R = B.copy()
    Pd = R.copy()
    rs = np.einsum("ij,ij->j", R, R)
    b2 = rs.copy()
    for it in range(maxit):
        AP = K @ Pd
        AP += delta * Pd
        alpha = rs / np.einsum("ij,ij->j", Pd, AP)
        X += alpha * Pd
        R -= alpha * AP
        rs_new = np.einsum("ij,ij->j", R, R)
        if np.all(rs_new <= tol * tol * b2):
            return X, it + 1
        Pd = R + (rs_new / rs) * Pd
        rs = rs_new
    raise AssertionError("conjugate gradients did not converge")


def test_eigen_free_route_beyond_the_syevd_index_range(ctx):
    from mixmogam_amd import linear_models as lm
    assert N > lm.EIGEN_FREE_MIN_N and N * N >= 2 ** 31
    T = {}
    t0 = time.time()
    g = ctx.geno(M=M, N=N).fill_hash(4711)
    # ---- exact GRM of one 50,000-SNP chunk (hdf5_data.py:99-106) in HBM, fetched once
    acc = ctx.kinship_accumulator(N)
    acc.add_grm(g)
    K, cnt = acc.fetch()
    acc.close()
    T["kinship+fetch"] = time.time() - t0
    assert cnt == M and K.shape == (N, N)
    cols = np.r_[0:24, N - 24:N]                            # both ends of the index range
    sub = np.ascontiguousarray(g.download()[:, cols]).astype(np.float64)
    mean_all, std_all = g.snp_stats()
    rows = np.r_[0:8, M - 8:M]
    s_rows = g.download_rows(rows).astype(np.float64)
    assert np.allclose(mean_all[rows], s_rows.mean(1), rtol=0, atol=1e-14) and np.allclose(std_all[rows], s_rows.std(1), rtol=1e-13)
    zs = (sub - mean_all[:, None]) / std_all[:, None]       # every SNP standardised over ALL individuals (hdf5_data.py:101)
    ref = zs.T @ zs
    got = K[np.ix_(cols, cols)]
    assert np.max(np.abs(got - ref)) < 1e-9 * np.max(np.abs(ref)), "GRM corner (first / last 24 individuals)"
    stride = 1499
    assert np.array_equal(K[cols][:, ::stride], K[::stride][:, cols].T)       # symmetric across the whole index range
    K /= float(M)

    rng = np.random.RandomState(7)
    y = rng.standard_normal(N)
    X = np.ones((N, 1))
    # ---- the four likelihood sums (linear_models.py:794-810 in closed form) at three delta values
    t0 = time.time()
    reml = ctx.reml(K, X, y)
    deltas = np.array([0.5, 2.0, 2.0 * (1 + 1e-3), 2.0 * (1 - 1e-3)])
    s1, s2, s3, s4, sse = reml.sums(deltas)
    T["reml_create+band_reduction+4_deltas"] = time.time() - t0
    t0 = time.time()
    c1, c2, c3, c4, _ = reml.sums(deltas[1:2], route="chol")          # one Cholesky factorisation + inverse at this width
    T["one_delta_cholesky_route"] = time.time() - t0
    for got, want in ((s1[1], c1[0]), (s2[1], c2[0]), (s3[1], c3[0]), (s4[1], c4[0])):
        assert abs(got - want) <= 1e-9 * max(1.0, abs(want)), ("band vs cholesky route", got, want)
    assert abs(sse - float(y @ y - y.sum() ** 2 / N)) < 1e-9 * sse
    t0 = time.time()
    B = np.column_stack([np.ones(N), y])
    its = []
    for k in (0, 1):
        U, it = _cg(K, deltas[k], B)
        its.append(it)
        a = float(U[:, 0].sum())                             # 1'H^-1 1
        Py = U[:, 1] - U[:, 0] * (float(U[:, 1].sum()) / a)  # P y = H^-1 y - H^-1 1 (1'H^-1 y) / (1'H^-1 1)
        assert abs(s1[k] / float(y @ Py) - 1) < 1e-8, ("s1 = y'Py", k)
        assert abs(s3[k] / float(Py @ Py) - 1) < 1e-8, ("s3 = |Py|^2", k)
    T["host_cg_sums"] = time.time() - t0
    # s4 = tr P = d s2 / d delta (s2 = log|H| + log|X'H^-1 X| - log|X'X|): central difference over +-0.1 %
    fd = (s2[2] - s2[3]) / (deltas[2] - deltas[3])
    assert abs(fd / s4[1] - 1) < 1e-5, "s4 against the derivative of s2"
    assert s2[0] < s2[1] and s4[0] > s4[1] > 0

    # ---- scan model on the device from K and delta (mmg_reml_scan_model), scan, sampled p-values in float64
    t0 = time.time()
    delta = 1.0
    h0_rss, beta = reml.scan_model(delta)
    reml.close()
    out = ctx.scan(g, h0_rss, N - 2, stats=True)
    T["scan_model+scan"] = time.time() - t0
    T["scan_quad_ms"] = ctx.kernel_ms("scan_quad")
    ps = out["ps"]
    assert ps.shape == (M,) and np.all(np.isfinite(ps)) and ps.min() > 0 and ps.max() <= 1
    sample = np.unique(np.r_[np.argsort(ps)[:8], rng.choice(M, 16, replace=False)])
    S = g.download_rows(sample).astype(np.float64).T        # N x 24
    t0 = time.time()
    U, it = _cg(K, delta, np.column_stack([np.ones(N), y, S]))
    T["host_cg_pvalues"] = time.time() - t0
    a = float(U[:, 0].sum())
    proj = lambda V: V - np.outer(U[:, 0], (np.ones(N) @ V) / a)     # H^-1 V -> P V for X = 1:  P = H^-1 - H^-1 1 1'H^-1 / (1'H^-1 1)
    Py = proj(U[:, 1:2])[:, 0]
    PS = proj(U[:, 2:])
    h0 = float(y @ Py)
    assert abs(h0_rss / h0 - 1) < 1e-9
    assert abs(float(beta[0]) / (float(U[:, 1].sum()) / a) - 1) < 1e-8   # GLS intercept (1'H^-1 y) / (1'H^-1 1)
    den = np.einsum("ij,ij->j", S, PS)
    dot = S.T @ Py
    rss = h0 - dot * dot / den
    F = (h0 / rss - 1.0) * (N - 2)
    want = orc.f_sf(F, 1, N - 2)
    assert np.max(np.abs(out["den"][sample] / den - 1)) < 1e-7
    assert np.max(np.abs(out["dot"][sample] / dot - 1)) < 1e-7 * max(1.0, np.max(np.abs(dot)) / np.min(np.abs(dot)))
    worst = float(np.max(np.abs(ps[sample] / want - 1)))
    print("N=%d M=%d: timings %s, CG iterations %s/%d, adaptive %s, max rel p err on %d sampled SNPs %.2e (min p %.2e)"
          % (N, M, {k: round(v, 2) for k, v in T.items()}, its, it, ctx.scan_last_stats(), len(sample), worst, ps.min()))
    assert worst < 1e-6
    g.close()

"""GPU parity tests (-m gpu) added in round 3: GRM weights as non-negative digits (negative genotype codes), the
upload roll-back, `return_transformed_snps`, the public permutation test against the reference's own numbers.
Same bars as the earlier files: bit-exact for integers, p-values within 1e-6 relative of the double-promoted
reference, tolerance written at each assert."""
import numpy as np
import pytest

from conftest import load_case, load_extras2

pytestmark = pytest.mark.gpu

orc = pytest.importorskip("oracle.emmax_oracle")


@pytest.fixture(scope="module")
def ctx():
    from mixmogam_amd import _lib
    return _lib.get_context()


@pytest.fixture(scope="module")
def ex2():
    return load_extras2()


def rel(a, b):
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))) if len(a) else 0.0


def _grm_f64(snps):
    s = snps.astype(np.float64)
    z = (s - s.mean(1, keepdims=True)) / s.std(1)[:, None]
    return z.T @ z


# ------------------------------------------------------------------ GRM: non-negative weight digits
@pytest.mark.parametrize("lo,hi,n,m", [(-1, 2, 300, 2000), (-1, 1, 257, 1500), (-2, 3, 130, 900), (-4, 5, 100, 700),
                                       (0, 2, 200, 70000)])
def test_grm_exact_route_with_negative_codes(ctx, lo, hi, n, m):
    """Genotype codes below zero (-1/0/1, -1/1 with 0 left out, -2..2, -4..4) through mmg_kin_acc_add_grm: the weight
    digits are non-negative and 7 / 6 / 5 bits wide, so digit * s fits int8 for either sign (round 2's balanced
    digits wrapped at -128 * -1, advisor finding).  The last case has >= 2^16 binary SNPs: 4 planes of 7 bits."""
    rng = np.random.RandomState(100 * (hi - lo) + n)
    snps = rng.randint(lo, hi, size=(m, n)).astype(np.int8)
    if (lo, hi) == (-1, 1):
        snps = (2 * rng.randint(0, 2, size=(m, n)) - 1).astype(np.int8)      # -1 / +1 only
    snps = snps[snps.std(1) > 0]
    ref = _grm_f64(snps)
    g = ctx.geno(snps)
    acc = ctx.kinship_accumulator(n)
    acc.add_grm(g)
    k1, cnt = acc.fetch()
    assert cnt == len(snps)
    # 28-30 bits of the largest weight per SNP, independent roundings: 1e-9 of the largest entry
    assert np.max(np.abs(k1 - ref)) < 1e-9 * np.max(np.abs(ref))
    acc.close(); g.close()


def test_grm_vs_golden_with_explicit_plane_counts(ctx, monkeypatch):
    """The double-promoted reference's calc_ibd_kinship (kinship.py:59-75) with 4, 5 and 6 weight planes."""
    from mixmogam_amd import kinship
    case = load_case("struct_n300_s2")
    for planes, tol in (("4", 2e-8), ("5", 3e-9), ("6", 3e-9)):
        monkeypatch.setenv("MMG_GRM_PLANES", planes)
        k = kinship.calc_ibd_kinship(case["snps"], ctx=ctx)
        assert np.max(np.abs(k - case["dbl_ibd_scaled"])) < tol, planes
    monkeypatch.delenv("MMG_GRM_PLANES")


# ------------------------------------------------------------------ upload roll-back
def test_upload_of_minus_128_rolls_back(ctx):
    """-128 has no negation in int8: the upload fails, the rows it wrote read as zeros, and the store's running
    bounds are what they were -- a later valid upload into the same store succeeds (advisor r2)."""
    import ctypes as C
    from mixmogam_amd import _lib
    n, m = 70, 40
    rng = np.random.RandomState(3)
    good = rng.randint(0, 3, size=(m, n)).astype(np.int8)
    g = ctx.geno(M=m, N=n)
    g.upload(good[:20], 0)
    bad = good[20:].copy()
    bad[5, 7] = -128
    with pytest.raises(_lib.MixmogamHipError, match="-128"):
        g.upload(bad, 20)
    assert np.array_equal(g.download()[:20], good[:20]) and not g.download()[20:].any()
    rc = ctx.lib.mmg_geno_upload(ctx.h, g.h, bad.ctypes.data_as(C.c_void_p), 20, len(bad))   # raw C ABI
    assert rc == -1 and b"-128" in ctx.lib.mmg_last_error(ctx.h)
    got = g.download()
    assert np.array_equal(got[:20], good[:20]) and not got[20:].any()
    g.upload(good[20:], 20)                                               # the bounds were restored
    assert np.array_equal(g.download(), good)
    # the linear-row fast path needs a 0/1 store: the failed upload must not have spoilt the tracked maximum
    acc = ctx.kinship_accumulator(n)
    acc.add_grm(g)
    k, cnt = acc.fetch()
    keep = good.std(1) > 0
    assert cnt == m and keep.all()
    assert np.max(np.abs(k - _grm_f64(good))) < 1e-9 * np.max(np.abs(k))
    acc.close(); g.close()
    assert isinstance(_lib.device_count(), int) and _lib.device_count() >= 1


# ------------------------------------------------------------------ return_transformed_snps
@pytest.mark.parametrize("tag", ["t", "tc"])
def test_return_transformed_snps_vs_golden(ctx, ex2, tag):
    """t_snps of _emmax_f_test_(return_transformed_snps=True) (linear_models.py:1309-1321,1355-1356) from the
    rotation GEMM with the rows of (I - QQ')H as its vectors: every row is digitised to 2^-27 of its largest entry, so
    an entry of t is a sum of <= N such roundings (sigma = 2^-28 max|row| sqrt(sum s^2 / 3)) -- 1e-7 of the largest |t|
    is > 8 sigma."""
    from mixmogam_amd import linear_models as lm
    n = int(ex2["n"])
    lmm = lm.LinearMixedModel(list(ex2["y"]), ctx=ctx)
    lmm.add_random_effect(ex2["ibs_scaled"])
    if tag == "tc":
        lmm.add_factor(ex2["cof"])
    r = lmm._emmax_f_test_(list(ex2["snps"][:64]), ex2["dbl_%s_H" % tag], return_transformed_snps=True, emma_num=0)
    ref = ex2["dbl_%s_snps" % tag]
    assert isinstance(r["t_snps"], list) and len(r["t_snps"]) == 64 and r["t_snps"][0].shape == (n,)
    assert np.max(np.abs(np.asarray(r["t_snps"]) - ref)) < 1e-7 * np.max(np.abs(ref))
    assert rel(r["ps"], ex2["dbl_%s_ps" % tag]) < 1e-6
    # the regression the reference runs on them (:1328): rss = |r|^2 - (t.r)^2 / (t.t) reproduces the scan's rss
    H = ex2["dbl_%s_H" % tag]
    prep = lmm.scan_prepare(H)
    t = np.asarray(r["t_snps"])
    rss = prep["h0_rss"] - (t @ prep["r"]) ** 2 / np.einsum("ij,ij->i", t, t)
    assert rel(rss, r["rss"]) < 1e-6


# ------------------------------------------------------------------ public permutation test
def test_emmax_perm_test_vs_the_reference(ctx, ex2):
    """emmax_perm_test -> LinearMixedModel.emmax_permutations (linear_models.py:1819-1841, :1180-1230) against the
    reference's own min_ps / max_f_stats (recorded shuffles; H_sqrt_inv as the reference computed it: its row signs
    are LAPACK's).  reference_indexing=True is the reference's literal per-SNP slotting (:1213)."""
    from mixmogam_amd import linear_models as lm
    k = int(ex2["perm_num_snps"])
    P = len(ex2["dbl_pub_perm_idx"])
    res = lm.emmax_perm_test(list(ex2["snps"][:k]), list(ex2["y"]), ex2["ibs_scaled"], num_perm=P,
                             perm_idx=ex2["dbl_pub_perm_idx"], H_sqrt_inv=ex2["dbl_pub_perm_H"],
                             reference_indexing=True, ctx=ctx)
    assert rel(res["max_f_stats"][:k], ex2["dbl_pub_max_f_stats"][:k]) < 1e-6
    assert np.max(np.abs(res["max_f_stats"][k:])) < 1e-9 and np.all(ex2["dbl_pub_max_f_stats"][k:] == 0)
    assert rel(res["min_ps"], ex2["dbl_pub_min_ps"]) < 1e-6
    p_f = sorted(zip(ex2["dbl_pub_min_ps"], ex2["dbl_pub_max_f_stats"]))
    assert rel(res["threshold_05"], p_f[len(p_f) // 20]) < 1e-6            # :1831
    with pytest.raises(IndexError):
        lm.emmax_perm_test(list(ex2["snps"][:40]), list(ex2["y"]), ex2["ibs_scaled"], num_perm=P,
                           perm_idx=ex2["dbl_pub_perm_idx"], H_sqrt_inv=ex2["dbl_pub_perm_H"],
                           reference_indexing=True, ctx=ctx)
    # the default reduction (min over the SNPs per permutation) against the oracle's restatement, all 500 SNPs
    n = int(ex2["n"])
    res2 = lm.emmax_perm_test(ex2["snps"], list(ex2["y"]), ex2["ibs_scaled"], num_perm=P,
                              perm_idx=ex2["dbl_pub_perm_idx"], H_sqrt_inv=ex2["dbl_pub_perm_H"], ctx=ctx)
    want = orc.perm_public(ex2["snps"], ex2["y"], np.ones((n, 1)), ex2["dbl_pub_perm_H"], ex2["dbl_pub_perm_idx"],
                           reference_indexing=False)
    assert rel(res2["max_f_stats"], want["max_f_stats"]) < 1e-6
    assert rel(res2["min_ps"], want["min_ps"]) < 1e-6
    # without a supplied H the wrapper computes its own (device eigh): the statistics stay in range
    res3 = lm.emmax_perm_test(ex2["snps"], list(ex2["y"]), ex2["ibs_scaled"], num_perm=10, ctx=ctx)
    assert len(res3["min_ps"]) == 10 and np.all((res3["min_ps"] > 0) & (res3["min_ps"] <= 1))


def test_uncentred_perm_plan_refuses_the_after_scan_form(ctx):
    from mixmogam_amd import _lib
    rng = np.random.RandomState(0)
    n = 64
    H = rng.standard_normal((n, n)) * 0.1 + np.eye(n)
    Ys = rng.standard_normal((n, 8))
    snps = rng.randint(0, 2, size=(100, n)).astype(np.int8)
    g = ctx.geno(snps)
    plan = ctx.perm_plan(H, Ys, 50.0, centre_snps=False)
    got = plan.run(g)
    T = snps.astype(np.float64) @ H.T
    want = np.minimum(50.0, np.einsum("ij,ij->j", Ys, Ys) - ((T @ Ys) ** 2 / np.einsum("ij,ij->i", T, T)[:, None]).max(0))
    assert rel(got, want) < 1e-7
    with pytest.raises(_lib.MixmogamHipError, match="centred"):
        plan.run(g, after_scan_HtQ=np.ones((1, n)))
    plan.close(); g.close()


# ------------------------------------------------------------------ packed genotype ingest
@pytest.mark.parametrize("bits,n,m", [(1, 61, 300), (1, 256, 513), (1, 5000, 700), (2, 61, 300), (2, 257, 1000),
                                      (2, 4999, 600)])
def test_packed_upload_round_trip_bit_exact(ctx, bits, n, m):
    """mmg_geno_upload_packed: 1 / 2 bits per genotype on the host, expanded to the int8 store on the device -- the
    downloaded rows equal the unpacked ones bit for bit, ragged last bytes, padding columns and row offsets included;
    a 4-entry lut re-codes on the fly (a PLINK .bed row: 00 hom A1, 01 missing, 10 het, 11 hom A2)."""
    from mixmogam_amd import _lib
    rng = np.random.RandomState(bits * 1000 + n)
    snps = rng.randint(0, 1 << bits, size=(m, n)).astype(np.int8)
    packed = _lib.pack_genotypes(snps, bits)
    g = ctx.geno(M=m + 7, N=n)
    g.upload(np.full((m + 7, n), 1, dtype=np.int8))                  # stale content everywhere
    g.upload_packed(packed, bits, m0=3)
    got = g.download()
    assert np.array_equal(got[3:3 + m], snps)
    assert np.all(got[:3] == 1) and np.all(got[3 + m:] == 1)          # rows outside [m0, m0 + rows) untouched
    # a wider host stride than ceil(N*bits/8)
    wide = np.zeros((m, packed.shape[1] + 5), dtype=np.uint8)
    wide[:, :packed.shape[1]] = packed
    wide[:, packed.shape[1]:] = 0xff                                   # junk beyond the row's last genotype
    g.upload_packed(wide, bits, m0=0)
    assert np.array_equal(g.download(0, m), snps)
    if bits == 2:
        lut = np.array([0, 1, 1, 2], dtype=np.int8)                    # .bed codes with "missing" imputed as 1
        g.upload_packed(packed, bits, m0=0, lut=lut)
        assert np.array_equal(g.download(0, m), lut[snps])
        with pytest.raises(_lib.MixmogamHipError):
            g.upload_packed(packed, bits, m0=0, lut=np.array([0, -128, 1, 2], dtype=np.int8))
    with pytest.raises(ValueError):
        g.upload_packed(packed[:, :-1], bits)
    g.close()
    # the padding columns of a REUSED store are rewritten: kinship counts of a packed upload == those of the int8 one
    g1, g2 = ctx.geno(snps), ctx.geno(M=m, N=n)
    g2.upload(np.full((m, n), 3, dtype=np.int8))
    g2.upload_packed(packed, bits)
    if bits == 1:
        assert np.array_equal(ctx.kinship_ibs_counts(g1), ctx.kinship_ibs_counts(g2))
        assert np.array_equal(ctx.kinship_ibs_counts(g2), orc.ibs_counts(snps))
    g1.close(); g2.close()


def test_run_emmax_over_a_packed_container_equals_the_int8_one(ctx, tmp_path):
    """hdf5_data.run_emmax streaming `raw_snps_packed` chunks (read, uploaded packed, expanded on the device, both
    passes double buffered) gives the p-values and the kinship of the int8 container bit for bit."""
    from mixmogam_amd import chunkstore, hdf5_data
    rng = np.random.RandomState(12)
    n = 333
    pops = rng.randint(0, 3, size=n)
    snps = {}
    for c in (1, 2, 3):
        fr = rng.uniform(0.15, 0.85, size=(900, 3))
        snps["chrom_%d" % c] = (rng.random_sample((900, n)) < fr[:, pops]).astype(np.int8)
    y = rng.randn(n) + 0.8 * snps["chrom_2"][17]
    a = chunkstore.write_genotype_container(str(tmp_path / "plain.mmg"), snps, range(n), phenotypes=y)
    b = chunkstore.write_genotype_container(str(tmp_path / "packed.mmg"), snps, range(n), phenotypes=y, packed_bits=1)
    ra = hdf5_data.run_emmax(a, None, min_maf=0.1, chunk_size=256, ctx=ctx)
    rb = hdf5_data.run_emmax(b, None, min_maf=0.1, chunk_size=256, ctx=ctx)
    assert ra["num_snps"] == rb["num_snps"]
    assert np.array_equal(ra["kinship"], rb["kinship"])
    for c in ra["chrom_results"]:
        assert np.array_equal(ra["chrom_results"][c]["ps"], rb["chrom_results"][c]["ps"])
    hdf5_data.release_pools()


def test_scan_model_falls_back_to_H_when_the_cholesky_route_fails(ctx, monkeypatch):
    """mlmm() and the chunked drivers build a step's scan model on the device from K and delta (Cholesky of K + delta
    I) but hold H_sqrt_inv as well: when the factorisation reports 'not positive definite' (an indefinite or
    numerically borderline kinship), _emmax_f_test_ builds the model from H_sqrt_inv instead of raising (advisor r2).
    The failure is injected; other device errors still propagate."""
    from mixmogam_amd import _lib, linear_models as lm
    case = load_case("struct_n150_s0")
    lmm = lm.LinearMixedModel(list(case["y"]), ctx=ctx)
    lmm.add_random_effect(case["dbl_ibs_scaled"])
    est = lmm.get_estimates(lmm._get_eigen_L_(), method="REML")
    want = lmm._emmax_f_test_(case["snps"], est["H_sqrt_inv"], emma_num=0)
    via_device = lmm._emmax_f_test_(case["snps"], est["H_sqrt_inv"], emma_num=0, _delta=est["delta"])
    assert rel(via_device["ps"], want["ps"]) < 1e-6

    def broken(self, delta, ndigits=0):
        raise _lib.MixmogamHipError("libmixmogam_hip error -4: K + delta I is not positive definite (dpotrf info 7)")
    monkeypatch.setattr(_lib.Reml, "scan_model", broken)
    got = lmm._emmax_f_test_(case["snps"], est["H_sqrt_inv"], emma_num=0, _delta=est["delta"])
    assert np.array_equal(got["ps"], want["ps"])
    with pytest.raises(_lib.MixmogamHipError):                      # nothing to fall back to
        lmm._emmax_f_test_(case["snps"], None, emma_num=0, _delta=est["delta"])

    def other(self, delta, ndigits=0):
        raise _lib.MixmogamHipError("libmixmogam_hip error -2: hipMalloc failed")
    monkeypatch.setattr(_lib.Reml, "scan_model", other)
    with pytest.raises(_lib.MixmogamHipError, match="hipMalloc"):
        lmm._emmax_f_test_(case["snps"], est["H_sqrt_inv"], emma_num=0, _delta=est["delta"])


# ------------------------------------------------------------------ indicator kinship on FP4 operands
@pytest.mark.parametrize("lo,hi,n,m", [(0, 3, 1000, 70000), (-1, 3, 333, 5000), (0, 2, 1100, 30000)])
def test_indicator_counts_on_fp4_operands_bit_exact(ctx, lo, hi, n, m):
    """mmg_kinship_indicator_counts (the two products of the 'diploid_int' IBS kinship, kinship.py:33-41) forms the
    indicator [s >= thr] as E2M1 nibbles and multiplies on the FP4 MFMA: counts are integers below 2^24 per K chunk,
    so the result is bit-identical to the integer product -- for 0/1/2, for negative codes (never >= thr) and for a
    binary store at a threshold that selects nothing."""
    rng = np.random.RandomState(n + m)
    snps = rng.randint(lo, hi, size=(m, n)).astype(np.int8)
    g = ctx.geno(snps)
    for thr in (1, 2):
        u = (snps >= thr).astype(np.float32)
        want = (u.T @ u).astype(np.int64)                  # m < 2^24: exact in float32
        assert np.array_equal(ctx.kinship_indicator_counts(g, thr), want), thr
    g.close()


# ------------------------------------------------------------------ REML sums through one band reduction
def _reml_sums_f64(K, X, y, delta):
    """s1..s4 of include/mixmogam_hip.h (linear_models.py:794-810 in closed form), dense float64 on the host."""
    n = len(y)
    Hi = np.linalg.inv(K + delta * np.eye(n))
    HiX = Hi @ X
    a = X.T @ HiX
    P = Hi - HiX @ np.linalg.solve(a, HiX.T)
    Py = P @ y
    s2 = np.linalg.slogdet(K + delta * np.eye(n))[1] + np.linalg.slogdet(a)[1] - np.linalg.slogdet(X.T @ X)[1]
    return float(y @ Py), float(s2), float(Py @ Py), float(np.trace(P))


@pytest.mark.parametrize("n,q", [(40, 1), (65, 2), (66, 1), (128, 3), (129, 1), (191, 2), (300, 1), (777, 4), (2000, 2)])
def test_reml_sums_band_route_vs_float64_and_the_cholesky_route(ctx, n, q):
    """mmg_reml_sums_ex(route = BAND): K reduced once to bandwidth 64 (Householder panels + symmetric rank-2b updates),
    every delta from banded Cholesky / solves / the band of the inverse -- against dense float64 on the host and
    against the one-factorisation-per-delta route, at sizes around the band width and its multiples.  Tolerance 1e-9
    relative (both routes are backward stable in float64; the sums are O(n) large)."""
    rng = np.random.RandomState(n + q)
    m = 3 * n
    pop = rng.randint(0, 3, size=n)
    f = np.clip(0.5 + 0.25 * rng.standard_normal((m, 3)), 0.05, 0.95)
    S = (rng.random_sample((m, n)) < f[:, pop]).astype(np.float64)
    S = S[S.std(1) > 0]
    Z = (S - S.mean(1, keepdims=True)) / S.std(1, keepdims=True)
    K = Z.T @ Z / len(Z)
    X = np.column_stack([np.ones(n)] + [rng.standard_normal(n) for _ in range(q - 1)])
    y = rng.standard_normal(n) + 2.0 * Z[0]
    deltas = np.exp(np.linspace(-6, 6, 9))
    reml = ctx.reml(K, X, y)
    band = reml.sums(deltas, route="band")
    again = reml.sums(deltas[3:5], route="band")            # the reduction is kept: later calls only run the band kernels
    chol = reml.sums(deltas, route="chol")
    reml.close()
    for k, d in enumerate(deltas):
        want = _reml_sums_f64(K, X, y, d)
        for i in range(4):
            assert abs(band[i][k] - want[i]) <= 1e-9 * max(abs(want[i]), 1.0), (n, q, d, i, band[i][k], want[i])
            assert abs(band[i][k] - chol[i][k]) <= 1e-9 * max(abs(want[i]), 1.0), (n, q, d, i)
    for i in range(4):
        assert np.array_equal(again[i], band[i][3:5])        # deterministic: same kernels, same order
    assert band[4] == chol[4]


@pytest.mark.parametrize("env", [{"MMG_BAND_BS": "512"}, {"MMG_BAND_BS": "256"}, {"MMG_BAND_IMPL": "lib"}])
def test_reml_band_reduction_variants_agree(ctx, monkeypatch, env):
    """The reduction's block-row grid at a size where it has several rows (MMG_BAND_BS: products and updates touch only
    blocks on and below the diagonal of that grid) and the library-calls-only form (MMG_BAND_IMPL=lib: rocSOLVER geqrf /
    larft, symm, syr2k) give the sums of the default form (own panel QR, one block row at this size) to 1e-10 -- different
    orthogonal bases, same invariants."""
    rng = np.random.RandomState(11)
    n = 1500
    B = rng.standard_normal((n, 300))
    K = B @ B.T / 300 + 0.05 * np.diag(rng.random_sample(n))
    X = np.column_stack([np.ones(n), rng.standard_normal(n)])
    y = rng.standard_normal(n)
    deltas = [1e-3, 0.3, 40.0]
    reml = ctx.reml(K, X, y)
    ref = reml.sums(deltas, route="band")
    reml.close()
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    reml = ctx.reml(K, X, y)
    got = reml.sums(deltas, route="band")
    reml.close()
    for i in range(4):
        assert np.max(np.abs(got[i] - ref[i]) / np.maximum(np.abs(ref[i]), 1.0)) < 1e-10, (env, i)


def test_reml_band_route_reports_an_indefinite_matrix(ctx):
    from mixmogam_amd import _lib
    rng = np.random.RandomState(5)
    n = 300
    B = rng.standard_normal((n, 40))
    K = B @ B.T / 40 - 0.5 * np.outer(B[:, 0], B[:, 0]) / 40 * 8
    K = 0.5 * (K + K.T)
    assert np.linalg.eigvalsh(K).min() < -0.01
    reml = ctx.reml(K, np.ones((n, 1)), rng.standard_normal(n))
    for route in ("band", "chol"):
        with pytest.raises(_lib.MixmogamHipError, match="positive definite"):
            reml.sums([1e-3], route=route)
    s = reml.sums([50.0], route="band")                       # large enough a delta makes it definite again
    assert np.isfinite(s[0][0])
    reml.close()


@pytest.mark.parametrize("n", [5000, 8500])
def test_emmax_routes_agree_just_above_the_eigen_free_threshold(ctx, monkeypatch, n):
    """linear_models.emmax() at the headline N = 5,000 and at N = 8,500 (> EIGEN_FREE_MIN_N: REML through the band reduction,
    scan model from one Cholesky factorisation, no eigh) against the same call with the threshold raised (eigh of K, REML
    from eig_L, the reference's route :1233-1267): variance components to 1e-8, p-values to 1e-7 relative -- with a
    cofactor, so that q = 2 runs through the banded solves."""
    from mixmogam_amd import linear_models as lm
    m = 3000
    assert n > lm.EIGEN_FREE_MIN_N
    g = ctx.geno(M=m, N=n).fill_structured(77, npop=3)
    rows = g.download()
    acc = ctx.kinship_accumulator(n)
    acc.add_grm(g)
    K, cnt = acc.fetch()
    acc.close()
    K = K / cnt
    rng = np.random.RandomState(3)
    cof = rng.standard_normal(n)
    y = rng.standard_normal(n) + 0.4 * cof + 0.8 * rows[5] - 0.7 * rows[99] + 2.0 * (K @ rng.standard_normal(n)) / np.sqrt(n)
    a = lm.emmax(g, list(y), K, cofactors=[cof], ctx=ctx)
    assert a["timings"]["eig_L"] == 0.0                       # the eigendecomposition-free branch ran
    monkeypatch.setattr(lm, "EIGEN_FREE_MIN_N", 10 ** 9)
    b = lm.emmax(g, list(y), K, cofactors=[cof], ctx=ctx)
    assert b["timings"]["eig_L"] > 0.0
    for k in ("pseudo_heritability", "ve", "vg", "max_ll"):
        assert abs(a[k] - b[k]) <= 1e-8 * max(1.0, abs(b[k])), (k, a[k], b[k])
    assert rel(a["ps"], b["ps"]) < 1e-7
    assert rel(a["h0_rss"], b["h0_rss"]) < 1e-9
    g.close()


def test_reml_band_route_on_a_rank_deficient_kinship(ctx):
    """Fewer SNPs than individuals (K of rank m - 1 = 299 at n = 600) and the whole grid of get_estimates,
    delta = e^-10 .. e^10: K + delta I has condition 1e5 at the small end; the band route still meets 1e-9 against dense
    float64 (measured 1e-11 there, 1e-15 from delta = 0.1 up)."""
    rng = np.random.RandomState(1)
    n, m = 600, 300
    S = (rng.random_sample((m, n)) < 0.4).astype(np.float64)
    Z = (S - S.mean(1, keepdims=True)) / S.std(1, keepdims=True)
    K = Z.T @ Z / m
    X = np.ones((n, 1))
    y = rng.standard_normal(n)
    deltas = np.exp(np.linspace(-10, 10, 11))
    reml = ctx.reml(K, X, y)
    band = reml.sums(deltas, route="band")
    reml.close()
    for k, d in enumerate(deltas):
        want = _reml_sums_f64(K, X, y, d)
        for i in range(4):
            assert abs(band[i][k] - want[i]) <= 1e-9 * max(abs(want[i]), 1.0), (d, i, band[i][k], want[i])


def test_scale_k_on_the_device_equals_the_host_rule(ctx):
    """mmg_kin_acc_scale_k: kinship.py:94-100 applied to the accumulated sum in HBM == kinship.scale_k(K / n_snps) on the
    host (the rule is invariant under the division); 1e-13: summation order only."""
    from mixmogam_amd import kinship
    rng = np.random.RandomState(2)
    for n, m in ((300, 4000), (2500, 3000)):                 # either branch of the host rule (n <= 2048 / beyond)
        snps = (rng.random_sample((m, n)) < rng.uniform(0.1, 0.9, size=(m, 1))).astype(np.int8)
        snps = snps[snps.std(1) > 0]
        g = ctx.geno(snps)
        acc = ctx.kinship_accumulator(n)
        acc.add_grm(g)
        raw, cnt = acc.fetch()
        want = kinship.scale_k(raw / float(cnt))
        f = acc.scale_k()
        got, _ = acc.fetch()
        acc.close(); g.close()
        assert np.max(np.abs(got - want)) <= 1e-13 * np.max(np.abs(want)), n
        assert abs(f * cnt / ((n - 1) / (np.trace(raw / cnt) - (raw / cnt).sum() / n)) - 1) < 1e-12


def test_fused_grm_over_several_contraction_chunks(ctx, monkeypatch):
    """The one-pass four-plane GRM (gemm_i8_grm4.h) when a call is split into several SNP ranges (MMG_KIN_CHUNK): each
    range brings its own digit rows and weighted column sums; N is not a multiple of the 128-column tile, M not of the
    128-row K step."""
    rng = np.random.RandomState(42)
    n, m = 333, 70001
    snps = (rng.random_sample((m, n)) < rng.uniform(0.08, 0.92, size=(m, 1))).astype(np.int8)
    snps = snps[snps.std(1) > 0]
    assert len(snps) >= 65536
    ref = _grm_f64(snps)
    monkeypatch.setenv("MMG_KIN_CHUNK", "20000")
    g = ctx.geno(snps)
    acc = ctx.kinship_accumulator(n)
    acc.add_grm(g)
    k1, cnt = acc.fetch()
    acc.close(); g.close()
    assert cnt == len(snps)
    assert np.max(np.abs(k1 - ref)) < 1e-9 * np.max(np.abs(ref))


def test_reml_sums_route_argument_is_checked(ctx):
    """mmg_reml_sums_ex refuses a route outside {AUTO, CHOL, BAND} with MMG_E_ARG; Reml.uses_band mirrors the AUTO rule."""
    from mixmogam_amd import _lib
    import ctypes as C
    rng = np.random.RandomState(0)
    n = 300
    B = rng.standard_normal((n, 50))
    K = B @ B.T / 50 + 0.1 * np.eye(n)
    reml = ctx.reml(K, np.ones((n, 1)), rng.standard_normal(n))
    d = np.array([1.0])
    out = [np.empty(1) for _ in range(4)]
    sse = C.c_double(0.0)
    rc = ctx.lib.mmg_reml_sums_ex(ctx.h, reml.h, 1, _lib._ptr(d), *[_lib._ptr(o) for o in out], C.byref(sse), 7)
    assert rc != 0
    assert reml.uses_band("band") and not reml.uses_band("chol") and reml.uses_band("auto")       # n = 300 >= 256
    a = reml.sums(d)                                            # AUTO = band here
    b = reml.sums(d, route="band")
    assert all(np.array_equal(x, y) for x, y in zip(a[:4], b[:4]))
    reml.close()

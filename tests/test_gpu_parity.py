"""GPU parity tests (-m gpu): the HIP path, called through the C ABI (ctypes) and through the
reference-shaped Python surface, against the CPU oracle and the committed golden vectors.

Bars: integer work (IBS counts, genotype store, hash fill) bit-exact; floating point within the
tolerance written next to each assert (north star: p-values 1e-6 relative vs the double-promoted
reference)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_case

pytestmark = pytest.mark.gpu

orc = pytest.importorskip("oracle.emmax_oracle")


@pytest.fixture(scope="module")
def ctx():
    from mixmogam_amd import _lib
    return _lib.get_context()          # raises (does not skip) when the .so or the GPU is missing


def rel(a, b):
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))) if len(a) else 0.0


def struct_snps(rng, n, m, npop=3):
    pops = rng.randint(0, npop, size=n)
    freqs = rng.uniform(0.05, 0.95, size=(m, npop))
    return (rng.random_sample((m, n)) < freqs[:, pops]).astype(np.int8)


# ------------------------------------------------------------------ genotype store
@pytest.mark.parametrize("n,m", [(150, 600), (257, 1000), (16, 3), (1000, 5)])
def test_geno_roundtrip(ctx, n, m):
    rng = np.random.RandomState(n + m)
    snps = rng.randint(0, 3, size=(m, n)).astype(np.int8)
    g = ctx.geno(snps)
    assert np.array_equal(g.download(), snps)
    assert np.array_equal(g.download(1, m - 1), snps[1:])
    g2 = ctx.geno(M=m, N=n)
    g2.upload(snps.astype(np.float32))
    assert np.array_equal(g2.download(), snps)
    g2.upload(snps[::-1].astype(np.float64))
    assert np.array_equal(g2.download(), snps[::-1])
    mean, sd = g.snp_stats()
    assert rel(mean, snps.mean(1)) < 1e-14
    assert np.max(np.abs(sd - snps.std(1))) < 1e-12


def test_fill_hash_matches_oracle(ctx):
    g = ctx.geno(M=777, N=333)
    g.fill_hash(20240, m_global0=1000, thr16=32768)
    assert np.array_equal(g.download(), orc.hash_genotypes(1000, 1777, 333, 20240))
    g.fill_hash(7, m_global0=0, thr16=9000)
    ref = orc.hash_genotypes(0, 777, 333, 7, maf_q16=np.full(777, 9000))
    assert np.array_equal(g.download(), ref)


# ------------------------------------------------------------------ kinship
def test_ibs_counts_bit_exact_golden(ctx, case):
    g = ctx.geno(case["snps"])
    c = ctx.kinship_ibs_counts(g)
    assert c.dtype == np.int64
    assert np.array_equal(c, case["dbl_ibs_counts"].astype(np.int64))


@pytest.mark.parametrize("n,m,seed", [(700, 5000, 0), (513, 12345, 1), (64, 100, 2), (1030, 257, 3)])
def test_ibs_counts_bit_exact_random(ctx, n, m, seed):
    rng = np.random.RandomState(seed)
    snps = struct_snps(rng, n, m)
    g = ctx.geno(snps)
    c = ctx.kinship_ibs_counts(g)
    ref = orc.ibs_counts(snps)
    bad = np.argwhere(c != ref)
    assert len(bad) == 0, "first mismatches %r got %r want %r" % (bad[:5], c[tuple(bad[0])], ref[tuple(bad[0])])
    # same counts through the fp32-MFMA affine kernel (exact: integers < 2^24)
    cf = ctx.kinship_affine(g)
    assert np.array_equal(cf, ref.astype(np.float64))


def test_kinship_chunked_snp_axis(ctx, monkeypatch):
    """Several passes over the SNP axis (what M > 4M triggers) give the same exact counts."""
    rng = np.random.RandomState(9)
    snps = struct_snps(rng, 300, 1000)
    g = ctx.geno(snps)
    ref = orc.ibs_counts(snps)
    monkeypatch.setenv("MMG_KIN_CHUNK", "256")
    assert np.array_equal(ctx.kinship_ibs_counts(g), ref)
    assert np.array_equal(ctx.kinship_affine(g), ref.astype(float))
    mean, sd = g.snp_stats()
    ok = sd > 0
    g2 = ctx.geno(snps[ok])
    z = (snps[ok] - mean[ok, None]) / sd[ok, None]
    got = ctx.kinship_affine(g2, 1.0 / sd[ok], -mean[ok] / sd[ok])
    assert np.max(np.abs(got - z.T @ z)) < 2e-3 * len(z) ** 0.5      # fp32 products, fp32 partial sums


def test_ibs_kinship_property_large(ctx):
    """size-independent properties at a size the CPU oracle would take minutes for:
    diagonal == M exactly, symmetry, and linearity in the SNP axis (counts of two halves add)."""
    n, m = 2000, 60000
    g = ctx.geno(M=m, N=n).fill_hash(11)
    c = ctx.kinship_ibs_counts(g)
    assert np.all(np.diag(c) == m)
    assert np.array_equal(c, c.T)
    g1 = ctx.geno(M=m // 2, N=n).fill_hash(11, m_global0=0)
    g2 = ctx.geno(M=m - m // 2, N=n).fill_hash(11, m_global0=m // 2)
    assert np.array_equal(ctx.kinship_ibs_counts(g1) + ctx.kinship_ibs_counts(g2), c)
    sub = orc.ibs_counts(g.download()[:, :64])
    assert np.array_equal(c[:64, :64], sub)


def test_kinship_module_golden(ctx, case):
    from mixmogam_amd import kinship
    k = kinship.calc_ibs_kinship(list(case["snps"]))
    assert rel(k, case["dbl_ibs_scaled"]) < 1e-13           # exact counts -> fp64 affine + scale_k
    ku = kinship.calc_ibs_kinship(case["snps"], scaled=False)
    assert np.array_equal(np.rint((ku - 0.5) * 2 * len(case["snps"])).astype(np.int64),
                          case["dbl_ibs_counts"].astype(np.int64))
    kd = kinship.calc_ibd_kinship(case["snps"])
    # fp32 MFMA accumulation of standardised genotypes: the reference's own fp32 accumulator
    # differs from its fp64 one by up to ~2e-7 absolute (golden lit vs dbl)
    assert np.max(np.abs(kd - case["dbl_ibd_scaled"])) < 2e-5
    assert rel(kinship.scale_k(ku), case["dbl_scale_k_of_ibs_unscaled"]) < 1e-13


# ------------------------------------------------------------------ dense fp64 helpers
def test_eigh_and_dgemm(ctx):
    rng = np.random.RandomState(3)
    n = 301
    a = rng.randn(n, n)
    a = a @ a.T / n + np.eye(n)
    vals, vecs = ctx.eigh(a)
    ref = np.linalg.eigvalsh(a)
    assert np.all(np.diff(vals) >= 0)                      # ascending: eigenvalue ORDER is exact
    assert np.max(np.abs(vals - ref)) < 1e-11
    assert np.max(np.abs(vecs @ vecs.T - np.eye(n))) < 1e-11
    assert np.max(np.abs((vecs.T * vals) @ vecs - a)) < 1e-10
    b = rng.randn(n, 77)
    c = rng.randn(55, n)
    assert rel(ctx.dgemm(a, b), a @ b) < 1e-12
    assert rel(ctx.dgemm(b, a, ta=True), b.T @ a) < 1e-12
    assert rel(ctx.dgemm(a, c, tb=True), a @ c.T) < 1e-12
    assert rel(ctx.dgemm(b, c, ta=True, tb=True), b.T @ c.T) < 1e-11


def test_f_sf_known_answers(ctx):
    kat = np.load(os.path.join(GOLDEN, "f_sf_kat.npz"))
    for nu in (197, 998, 4998, 49998):
        p = ctx.f_sf(kat["F"], nu)
        ref = kat["sf_%d" % nu]
        ok = ref > 1e-300
        # scipy itself loses ~1e-8 for F < 1e-3 (forms x = nu/(nu+F) before 1-x); 1e-7 covers it
        assert rel(p[ok], ref[ok]) < 1e-7, nu
        big = ok & (kat["F"] > 1e-3)
        assert rel(p[big], ref[big]) < 1e-9, nu
    assert ctx.f_sf([0.0, -1.0], 100).tolist() == [1.0, 1.0]


# ------------------------------------------------------------------ EMMAX scan
def _prep(case):
    y = case["y"]
    n = len(y)
    X = np.ones((n, 1))
    if case["cof"] is not None:
        X = np.hstack([X] + [c.reshape(n, 1) for c in case["cof"]])
    K = orc.scale_k(case["dbl_ibs_scaled"])
    est = orc.get_estimates(y, X, K)
    return orc.scan_prepare(y, X, est["H_sqrt_inv"])


@pytest.mark.parametrize("ndigits", [3, 4, 5, 6])
def test_scan_c_abi_vs_oracle(ctx, case, ndigits):
    prep = _prep(case)
    ref = orc.scan_closed(case["snps"], prep)
    g = ctx.geno(case["snps"])
    ctx.scan_set_model(prep["A"], prep["w"], ndigits)
    out = ctx.scan(g, prep["h0_rss"], prep["n"] - prep["q"] - 1, stats=True)
    # ndigits = number of 7-bit unsigned digit planes (gemm_i8_core.h SCAN_DIGIT_BITS): 3 planes = 20 bits of magnitude
    # relative to the largest off-diagonal entry (an explicitly reduced-precision mode), 4 = 27, 5 = 34
    # (6 planes: the offset term alone exceeds 2^63 at these sizes -- it is taken out in modular 64-bit arithmetic)
    tol = {3: 3e-4, 4: 1e-6, 5: 1e-6, 6: 1e-6}[ndigits]
    assert rel(out["dot"], case["snps"].astype(float) @ prep["w"]) < 1e-10
    assert rel(out["den"], ref["den"]) < tol * 1e-1
    assert rel(out["rss"], ref["rss"]) < tol * 1e-1
    assert rel(out["ps"], ref["ps"]) < tol, "min p %g" % ref["ps"].min()
    assert np.array_equal(out["sum"], case["snps"].sum(1).astype(float))


def test_scan_bitwise_reproducible_and_schedule_independent(ctx):
    case = load_case("struct_n300_s2")
    prep = _prep(case)
    g = ctx.geno(case["snps"])
    ctx.scan_set_model(prep["A"], prep["w"], 4)
    a = ctx.scan(g, prep["h0_rss"], 298)
    b = ctx.scan(g, prep["h0_rss"], 298)
    for k in ("rss", "f_stats", "ps"):
        assert np.array_equal(a[k], b[k])


def test_emmax_python_surface_vs_golden(ctx, case_emmax):
    """Every small case plus (round 6) the reference runs at N = 1000 (two cofactors) and in config 1's shape."""
    from mixmogam_amd import linear_models as lm
    case = case_emmax
    res = lm.emmax(list(case["snps"]), list(case["y"]), case["dbl_ibs_scaled"], cofactors=case["cof"])
    assert rel(res["ps"], case["dbl_emmax_ps"]) < 1e-6          # the north-star tolerance
    assert rel(res["rss"], case["dbl_emmax_rss"]) < 1e-8
    assert rel(res["h0_rss"], case["dbl_emmax_h0_rss"]) < 1e-9
    big = case["dbl_emmax_f_stats"] > 1e-6
    assert rel(res["f_stats"][big], case["dbl_emmax_f_stats"][big]) < 1e-6
    for k in ("pseudo_heritability", "ve", "vg", "max_ll"):
        assert rel(res[k], case["dbl_emmax_" + k]) < 1e-7, k
    assert np.argmin(res["ps"]) == np.argmin(case["dbl_emmax_ps"])


def test_reml_python_surface_vs_golden(ctx, case_reml):
    from mixmogam_amd import linear_models as lm
    case = case_reml
    res = lm.get_emma_reml_estimates(list(case["y"]), case["dbl_ibs_scaled"], cofactors=case["cof"])
    for k in ("max_ll", "delta", "ve", "vg", "pseudo_heritability"):
        assert rel(res[k], case["dbl_reml_" + k]) < 1e-7, k
    assert rel(res["beta"], case["dbl_reml_beta"]) < 1e-6
    ev = res["eig_L"]["values"]
    assert np.all(np.diff(ev) >= 0)
    assert np.max(np.abs(ev - case["dbl_eig_L_values"])) < 1e-9
    H = res["H_sqrt_inv"]
    probe = np.random.RandomState(99).randn(len(case["y"]), 3)
    assert rel(H.T @ (H @ probe), case["dbl_HtH_probe"]) < 1e-6


def test_with_betas_vs_golden(ctx, case):
    from mixmogam_amd import linear_models as lm
    lmm = lm.LinearMixedModel(list(case["y"]))
    lmm.add_random_effect(case["dbl_ibs_scaled"])
    if case["cof"] is not None:
        for c in case["cof"]:
            lmm.add_factor(c)
    wb = lmm.emmax_f_test(case["snps"][:200], with_betas=True, emma_num=0)
    assert rel(wb["ps"], case["dbl_wb_ps"]) < 1e-6
    got = np.asarray(wb["betas"])
    want = case["dbl_wb_betas"]
    assert got.shape == want.shape
    assert np.max(np.abs(got - want)) < 1e-6 * max(1.0, np.abs(want).max())


def test_scan_edge_cases(ctx):
    case = load_case("struct_n150_s0")
    prep = _prep(case)
    n = 150
    ctx.scan_set_model(prep["A"], prep["w"], 4)
    # monomorphic SNPs (all 0 / all 1): rss stays h0_rss, F = 0, p = 1 (linear_models.py:1308,1329)
    snps = np.vstack([np.zeros((1, n)), np.ones((1, n)), case["snps"][:3]]).astype(np.int8)
    out = ctx.scan(ctx.geno(snps), prep["h0_rss"], n - 2)
    assert out["rss"][0] == prep["h0_rss"] and out["rss"][1] == prep["h0_rss"]
    assert out["ps"][0] == 1.0 and out["ps"][1] == 1.0 and out["f_stats"][0] == 0.0
    ref = orc.scan_closed(case["snps"][:3], prep)
    assert rel(out["ps"][2:], ref["ps"]) < 1e-6
    # a single SNP and an empty store
    one = ctx.scan(ctx.geno(case["snps"][5:6]), prep["h0_rss"], n - 2)
    assert rel(one["ps"], orc.scan_closed(case["snps"][5:6], prep)["ps"]) < 1e-6
    empty = ctx.scan(ctx.geno(M=0, N=n), prep["h0_rss"], n - 2)
    assert len(empty["ps"]) == 0
    # allele-coding flip s -> 1 - s leaves F unchanged (the intercept is projected out)
    a = ctx.scan(ctx.geno(case["snps"]), prep["h0_rss"], n - 2)
    b = ctx.scan(ctx.geno(1 - case["snps"]), prep["h0_rss"], n - 2)
    assert rel(b["ps"], a["ps"]) < 1e-6
    # diploid-style 0/1/2 coding
    s2 = (case["snps"][:100] + case["snps"][100:200]).astype(np.int8)
    c = ctx.scan(ctx.geno(s2), prep["h0_rss"], n - 2)
    assert rel(c["ps"], orc.scan_closed(s2, prep)["ps"]) < 1e-6


def test_scan_midsize_vs_oracle(ctx):
    """N = 1100 (5 column tiles, ragged), M = 3000: full pipeline through the Python surface."""
    from mixmogam_amd import kinship, linear_models as lm
    rng = np.random.RandomState(42)
    n, m = 1100, 3000
    snps = struct_snps(rng, n, m)
    snps = snps[(snps.sum(1) > 0) & (snps.sum(1) < n)]
    y = snps[:8].astype(float).T @ rng.exponential(1.0, 8) + 2.0 * rng.randn(n)
    K = kinship.calc_ibs_kinship(snps)
    assert rel(K, orc.calc_ibs_kinship(snps)) < 1e-13
    res = lm.emmax(snps, y, K)
    ref = orc.emmax(snps, y, K)
    assert rel(res["ps"], ref["ps"]) < 1e-6, (res["ps"].min(), ref["ps"].min())
    assert rel(res["pseudo_heritability"], ref["pseudo_heritability"]) < 1e-6


def test_one_shot_c_abi(ctx):
    import ctypes as C
    from mixmogam_amd import _lib
    case = load_case("struct_n150_s0")
    prep = _prep(case)
    snps = np.ascontiguousarray(case["snps"])
    m, n = snps.shape
    rss, F, p = np.empty(m), np.empty(m), np.empty(m)
    A = np.ascontiguousarray(prep["A"])
    w = np.ascontiguousarray(prep["w"])
    rc = ctx.lib.mmg_emmax_scan_i8(ctx.h, _lib._ptr(snps), m, n, _lib._ptr(A), _lib._ptr(w),
                                   C.c_double(prep["h0_rss"]), n - 2, _lib._ptr(rss), _lib._ptr(F), _lib._ptr(p))
    assert rc == 0
    assert rel(p, orc.scan_closed(snps, prep)["ps"]) < 1e-6
    Cout = np.empty((n, n))
    rc = ctx.lib.mmg_kinship_i8(ctx.h, _lib._ptr(snps), m, n, None, None, _lib._ptr(Cout))
    assert rc == 0
    assert np.array_equal(Cout, orc.ibs_counts(snps).astype(float))
    # float32-genotype twins (the C3 config hands over fp32 [M x N] genotypes): same results
    s32 = snps.astype(np.float32)
    p32, C32 = np.empty(m), np.empty((n, n))
    assert ctx.lib.mmg_emmax_scan_f32(ctx.h, _lib._ptr(s32), m, n, _lib._ptr(A), _lib._ptr(w), C.c_double(prep["h0_rss"]),
                                      n - 2, None, None, _lib._ptr(p32)) == 0
    assert np.array_equal(p32, p)
    assert ctx.lib.mmg_kinship_f32(ctx.h, _lib._ptr(s32), m, n, None, None, _lib._ptr(C32)) == 0
    assert np.array_equal(C32, Cout)
    s32[3, 7] = 0.5                                                       # a dosage: refused, not rounded
    assert ctx.lib.mmg_kinship_f32(ctx.h, _lib._ptr(s32), m, n, None, None, _lib._ptr(C32)) != 0
    assert b"integers" in ctx.lib.mmg_last_error(ctx.h)
    # one-shot permutation test over host genotypes == the resident form
    H = np.ascontiguousarray(case["dbl_perm_H"])
    Ys = np.ascontiguousarray(prep["r"][case["dbl_perm_idx"]].T)
    mn = np.empty(Ys.shape[1])
    assert ctx.lib.mmg_emmax_perm_i8(ctx.h, _lib._ptr(snps), m, n, _lib._ptr(H), _lib._ptr(Ys), Ys.shape[1],
                                     C.c_double(prep["h0_rss"]), _lib._ptr(mn)) == 0
    gres = ctx.geno(snps)
    assert np.array_equal(mn, ctx.perm(gres, H, Ys, prep["h0_rss"]))
    gres.close()
    # error behaviour: scan before a model of matching N -> error code + message, no crash
    g = ctx.geno(M=4, N=n + 1)
    rc = ctx.lib.mmg_emmax_scan_device(ctx.h, g.h, C.c_double(1.0), 10)
    assert rc != 0 and b"N" in ctx.lib.mmg_last_error(ctx.h)


# ------------------------------------------------------------------ permutations
@pytest.mark.parametrize("name", ["struct_n150_s0", "struct_n300_s2", "bern_n200_s4", "struct_n1000_s7"])
def test_permutations_vs_golden_and_oracle(ctx, name, monkeypatch):
    from mixmogam_amd import linear_models as lm
    case = load_case(name)
    # The permutation test shuffles the ELEMENTS of the rotated residual H(y - Xb): its result
    # depends on the sign (and, for repeated eigenvalues, the basis) LAPACK happens to return for
    # each eigenvector -- scipy's eigh gives different signs on different CPUs.  H_sqrt_inv is an
    # argument of _emmax_permutations_ (linear_models.py:1125), so the fixture carries the one the
    # reference was run with; everything downstream is the device path.
    y, n = case["y"], len(case["y"])
    X = np.ones((n, 1))
    est = {"H_sqrt_inv": case["dbl_perm_H"]}
    lmm = lm.LinearMixedModel(list(case["y"]))
    lmm.add_random_effect(case["dbl_ibs_scaled"])
    res = lmm._emmax_permutations_(case["snps"], None, case["dbl_perm_H"], num_perm=len(case["dbl_perm_idx"]),
                                   perm_idx=case["dbl_perm_idx"])
    assert rel(res["max_f_stats"], case["dbl_perm_max_f_stats"]) < 1e-6
    assert rel(res["min_ps"], case["dbl_perm_min_ps"]) < 1e-5      # p ~ 1e-4: d ln p ~ 8 d ln F
    # the same through the raw C ABI against the oracle
    pp = orc.perm_prepare(y, X, est["H_sqrt_inv"], case["dbl_perm_idx"])
    ref = orc.perm_closed(case["snps"], pp)
    # t.t of the stand-alone test is the quadratic form of the centred model and follows the adaptive digit schedule
    # (within 2.5e-7 of itself by construction; min_rss moves by that times (G^2/tt)/rss, a few percent here); with every
    # plane for every SNP the agreement with float64 is that of the exact integers
    got = ctx.perm(ctx.geno(case["snps"]), est["H_sqrt_inv"], pp["Ys"], pp["h0_rss"])
    assert rel(got, ref["min_rss"]) < 2e-8
    monkeypatch.setenv("MMG_SCAN_ADAPTIVE", "0")
    got = ctx.perm(ctx.geno(case["snps"]), est["H_sqrt_inv"], pp["Ys"], pp["h0_rss"])
    monkeypatch.delenv("MMG_SCAN_ADAPTIVE")
    # round 3: the W digits of the permutation GEMM are 4 unsigned 7-bit planes (28 bits of each column's largest
    # entry; round 2: 31 bits in balanced base-256 digits, 1e-9 here) -- G carries ~2e-9 relative, min_rss a tenth of it
    assert rel(got, ref["min_rss"]) < 5e-9


def test_permutations_ragged_and_many(ctx):
    """P not a multiple of 64, M not a multiple of 256, N not a multiple of 256; includes a
    monomorphic SNP (constant after centring: contributes nothing)."""
    rng = np.random.RandomState(8)
    n, m, P = 333, 1500, 130
    snps = struct_snps(rng, n, m)
    snps[7] = 1
    y = rng.randn(n) + snps[3]
    X = np.ones((n, 1))
    K = orc.calc_ibs_kinship(snps)
    est = orc.get_estimates(y, X, orc.scale_k(K))
    idx = np.array([np.random.RandomState(100 + p).permutation(n) for p in range(P)])
    pp = orc.perm_prepare(y, X, est["H_sqrt_inv"], idx)
    keep = np.ones(m, bool)
    keep[7] = False
    ref = orc.perm_closed(snps[keep], pp)
    got = ctx.perm(ctx.geno(snps), est["H_sqrt_inv"], pp["Ys"], pp["h0_rss"])
    assert rel(got, ref["min_rss"]) < 5e-9                      # 28-bit W digits (round 3), see above
    # sharding property: min over two SNP halves == min over all (what the RCCL MIN all-reduce does)
    a = ctx.perm(ctx.geno(snps[:700]), est["H_sqrt_inv"], pp["Ys"], pp["h0_rss"])
    b = ctx.perm(ctx.geno(snps[700:]), est["H_sqrt_inv"], pp["Ys"], pp["h0_rss"])
    assert np.array_equal(np.minimum(a, b), got)


def test_emmax_perm_test_surface(ctx):
    from mixmogam_amd import linear_models as lm
    case = load_case("struct_n150_s0")
    res = lm.emmax_perm_test(case["snps"], list(case["y"]), case["dbl_ibs_scaled"], num_perm=40,
                             perm_idx=[np.random.RandomState(p).permutation(150) for p in range(40)])
    assert len(res["min_ps"]) == 40 and np.all(res["min_ps"] > 0) and np.all(res["min_ps"] <= 1)
    assert res["threshold_05"] == sorted(zip(res["min_ps"], res["max_f_stats"]))[2]


# ------------------------------------------------------------------ "next" rows
def test_linear_model_fast_f_test_vs_golden(ctx, case):
    """N4: LinearModel.fast_f_test / linear_model() = the scan with H = I."""
    from mixmogam_amd import linear_models as lm
    res = lm.linear_model(list(case["snps"]), list(case["y"]), cofactors=case["cof"])
    assert rel(res["h0_rss"], case["dbl_lm_h0_rss"]) < 1e-9
    assert rel(res["rss"], case["dbl_lm_rss"]) < 1e-7          # adaptive digit schedule: den to ~3e-8 below the F threshold
    assert rel(res["ps"], case["dbl_lm_ps"]) < 1e-6


@pytest.mark.parametrize("name", ["struct_n150_s0", "struct_n150_s1"])
def test_diploid_int_kinship_bit_exact(ctx, name):
    """N4: 'diploid_int' IBS kinship from two exact indicator GEMMs (kinship.py:33-41)."""
    from mixmogam_amd import kinship
    case = load_case(name)
    half = len(case["snps"]) // 2
    dip = (case["snps"][:half] + case["snps"][half:2 * half]).astype(np.int8)
    k = kinship.calc_ibs_kinship(list(dip), snps_data_format='diploid_int', scaled=False)
    assert np.array_equal(k, case["dbl_dip_ibs_unscaled"])
    g = ctx.geno(dip)
    for thr in (1, 2):
        u = (dip >= thr).astype(np.int64)
        assert np.array_equal(ctx.kinship_indicator_counts(g, thr), u.T @ u)


@pytest.mark.parametrize("name", ["struct_n150_s0", "struct_n150_s1"])
def test_exact_emma_refinement_vs_golden(ctx, name):
    """N2: emma_num > 0 -- the top hits are re-estimated with one device eigh per SNP."""
    from mixmogam_amd import linear_models as lm
    case = load_case(name)
    res = lm.emmax(case["snps"], list(case["y"]), case["dbl_ibs_scaled"], cofactors=case["cof"], emma_num=15)
    assert rel(res["ps"], case["dbl_emma15_ps"]) < 1e-5
    changed = np.argsort(case["dbl_emmax_ps"], kind="stable")[:15]
    assert rel(res["rss"][changed], case["dbl_emma15_rss"][changed]) < 1e-6
    assert rel(res["f_stats"][changed], case["dbl_emma15_f_stats"][changed]) < 1e-5


def test_snp_priors_bayes_factors(ctx):
    """N1 ingredient: snp_priors -> bfs / pos / ppas (linear_models.py:1311-1314,1357-1363)."""
    from mixmogam_amd import linear_models as lm
    case = load_case("struct_n150_s0")
    m = len(case["snps"])
    lmm = lm.LinearMixedModel(list(case["y"]))
    lmm.add_random_effect(case["dbl_ibs_scaled"])
    pri = np.full(m, 1.0 / m)
    res = lmm.emmax_f_test(case["snps"], snp_priors=pri, emma_num=0)
    n = 150
    h0 = float(res["h0_rss"][0])
    bfs = np.exp(((np.log(h0) - np.log(res["rss"])) * n - np.log(n)) / 2)
    assert rel(res["bfs"], bfs) < 1e-12
    assert rel(res["ppas"], (bfs * pri / (1 - pri)) / (1 + bfs * pri / (1 - pri))) < 1e-12


def test_examples_mixed_model_gwas_call_sequence(ctx, tmp_path):
    """BASELINE configs[0]: the examples.py:65-102 call sequence on the real FT10 phenotype (198 accessions)
    with synthetic genotypes, against the oracle."""
    import examples
    out = str(tmp_path / "mm.pvals")
    res, sd, phend, K = examples.mixed_model_gwas(pvalue_file=out, num_snps=20000)
    snps = np.asarray(sd.get_snps())
    y = np.asarray(phend.get_values(5))
    assert snps.shape[1] == 198 == len(y)
    assert rel(K, orc.calc_ibs_kinship(snps)) < 1e-13
    ref = orc.emmax(snps, y, K)
    assert rel(res["ps"], ref["ps"]) < 1e-6
    lines = open(out).read().splitlines()
    assert lines[0] == "chromosomes,positions,scores" and len(lines) == len(snps) + 1


def _chunk_source(rng, n, sizes):
    src, allsnps = {}, []
    for ci, m in enumerate(sizes):
        snps = struct_snps(rng, n, m)
        snps = snps[(snps.sum(1) > 0) & (snps.sum(1) < n)]
        src["chr%d" % (ci + 1)] = {"raw_snps": snps, "freqs": snps.mean(1), "positions": np.arange(len(snps)) * 10}
        allsnps.append(snps)
    return src, allsnps


def test_chunked_run_emmax_vs_oracle(ctx):
    """hdf5_data.run_emmax compute (MAF filter, GRM kinship over chunks, REML once, scan per chromosome)."""
    from mixmogam_amd import hdf5_data
    rng = np.random.RandomState(21)
    n = 260
    src, allsnps = _chunk_source(rng, n, [900, 700])
    y = allsnps[0][:6].astype(float).T @ rng.exponential(1.0, 6) + 2.0 * rng.randn(n)
    out = hdf5_data.run_emmax(src, y, min_maf=0.1, chunk_size=256, ctx=ctx)
    keep = [np.minimum(s.mean(1), 1 - s.mean(1)) > 0.1 for s in allsnps]
    filt = np.vstack([s[k] for s, k in zip(allsnps, keep)])
    assert out["num_snps"] == len(filt)
    kref = orc.calc_ibd_kinship(filt)
    assert np.max(np.abs(out["kinship"] - kref)) < 2e-5          # fp32 MFMA products
    ref = orc.emmax(filt, y, out["kinship"])                     # same K: isolates the chunked scan
    got = np.concatenate([out["chrom_results"][c]["ps"] for c in ("chr1", "chr2")])
    assert rel(got, ref["ps"]) < 1e-6
    assert rel(out["pseudo_heritability"], ref["pseudo_heritability"]) < 1e-6
    assert np.array_equal(out["chrom_results"]["chr2"]["positions"], src["chr2"]["positions"][keep[1]])
    # chunking is invisible: one big chunk gives the same p-values bit for bit given the same kinship
    out1 = hdf5_data.run_emmax(src, y, min_maf=0.1, chunk_size=10 ** 6, k=out["kinship"], ctx=ctx)
    got1 = np.concatenate([out1["chrom_results"][c]["ps"] for c in ("chr1", "chr2")])
    assert np.array_equal(got1, got)


def test_chunked_run_emmax_perm(ctx):
    from mixmogam_amd import hdf5_data, linear_models as lm
    rng = np.random.RandomState(22)
    n = 200
    src, allsnps = _chunk_source(rng, n, [500, 400])
    y = rng.randn(n) + allsnps[1][5]
    idx = [np.random.RandomState(p).permutation(n) for p in range(30)]
    a = hdf5_data.run_emmax_perm(src, y, min_maf=0.05, chunk_size=128, num_perm=30, perm_idx=idx, ctx=ctx)
    b = hdf5_data.run_emmax_perm(src, y, min_maf=0.05, chunk_size=10 ** 6, num_perm=30, perm_idx=idx,
                                 k=a["kinship"], ctx=ctx)
    assert rel(a["perm_min_ps"], b["perm_min_ps"]) < 1e-9        # min over chunks == min over all SNPs
    keep = [np.minimum(s.mean(1), 1 - s.mean(1)) > 0.05 for s in allsnps]
    # hdf5_data.py:294-311: the permutation test sees every chromosome but the LAST (`chr12_snps`)
    filt = np.vstack([s[k] for s, k in zip(allsnps, keep)][:-1])
    # round 5: the driver's H_sqrt_inv is L^-1 of K + delta I = L L' (no eigendecomposition); the oracle on the same matrix
    from mixmogam_amd import kinship
    delta = 1.0 / a["pseudo_heritability"] - 1.0
    H = np.linalg.inv(np.linalg.cholesky(kinship.scale_k(np.asarray(a["kinship"])) + delta * np.eye(n)))
    pp = orc.perm_prepare(y, np.ones((n, 1)), H, np.asarray(idx))
    ref = orc.perm_closed(filt, pp)
    assert rel(a["perm_max_f_stats"], ref["max_f_stats"]) < 1e-6
    assert a["threshold_05"][0] == np.sort(a["perm_min_ps"])[1]
    # ... and MMG_PERM_H=eigen keeps the literal route: the eigendecomposition's matrix (the same deterministic device eigenbasis)
    import os
    os.environ["MMG_PERM_H"] = "eigen"
    try:
        c = hdf5_data.run_emmax_perm(src, y, min_maf=0.05, chunk_size=128, num_perm=30, perm_idx=idx, k=a["kinship"], ctx=ctx)
    finally:
        del os.environ["MMG_PERM_H"]
    lmm = lm.LinearMixedModel(y, ctx=ctx)
    lmm.add_random_effect(a["kinship"])
    eL, eR = lmm._get_eigen_L_(), lmm._get_eigen_R_(X=lmm.X)
    He = lmm._get_estimates_with(eL, eR, "REML")["H_sqrt_inv"]
    refe = orc.perm_closed(filt, orc.perm_prepare(y, np.ones((n, 1)), He, np.asarray(idx)))
    assert rel(c["perm_max_f_stats"], refe["max_f_stats"]) < 1e-6


@pytest.mark.parametrize("name", ["struct_n150_s0", "struct_n1000_s7"])
def test_mlmm_forward_backward_vs_golden(ctx, name):
    """N1: multi-locus mixed model (forward inclusion + backward elimination), every step one device scan
    over the resident genotypes; step statistics vs the reference's own mlmm run (golden; round 6: also at N = 1000)."""
    from mixmogam_amd import linear_models as lm
    case = load_case(name)
    m = len(case["snps"])
    res = lm.mlmm(list(case["y"]), case["dbl_ibs_scaled"], num_steps=3, forward_backwards=True,
                  snps=case["snps"], positions=list(range(m)), chromosomes=[1] * m, ctx=ctx)
    keys = ("pseudo_heritability", "ll", "bic", "e_bic", "m_bic", "mbonf", "min_pval", "rss",
            "reml_mahalanobis_rss", "mahalanobis_rss")
    want = case["dbl_mlmm_steps"]
    assert len(res["step_info_list"]) == len(want)
    for si, row, pos, mlogp in zip(res["step_info_list"], want, case["dbl_mlmm_cof_pos"], case["dbl_mlmm_cof_mlogp"]):
        for k, w in zip(keys, row):
            if np.isnan(w):
                assert si[k] is None, k
            elif w == 0:
                assert float(np.asarray(si[k]).reshape(-1)[0]) == 0
            else:
                assert abs(float(np.asarray(si[k]).reshape(-1)[0]) / w - 1) < 2e-6, (k, si[k], w)
        assert [c[1] for c in si["cofactors"]] == [p for p in pos if p >= 0]
        for c, w in zip(si["cofactors"], mlogp):
            assert abs(c[2] / w - 1) < 1e-6
    for c in ("ebics", "mbics", "bonf", "mbonf", "min_cof_ppa"):
        assert res["opt_dict"][c] == int(case["dbl_mlmm_opt_" + c]), c
    if "dbl_mlmm_first_ps" in case:
        assert rel(res["first_emmax_res"]["ps"], case["dbl_mlmm_first_ps"]) < 1e-6


@pytest.mark.parametrize("variant", ["q8", "w4m", "w4b", "bits", "flat", "ring"])
def test_scan_kernel_generations_agree_bit_for_bit(ctx, monkeypatch, variant):
    """Every generation of the quadratic-form GEMM (MMG_SCAN_KERNEL) accumulates the same exact integers:
    den / rss / p are bit-identical to the production kernel on a ragged multi-tile problem, for binary
    genotypes (all kernels) and for 0/1/2 genotypes (where the bit-packed ones fall back).  The superseded
    generations live in csrc/experiments/ and are only in a `make EXPERIMENTS=1` library (run this test with
    MMG_LIB pointing at it); the shipped library has the production kernel only."""
    if not ctx.lib.mmg_has_experiments():
        pytest.skip("library built without csrc/experiments/")
    rng = np.random.RandomState(5)
    n, m = 1100, 2500
    B = rng.standard_normal((n, 30)) / 6
    A = np.eye(n) + B @ B.T / n
    w = rng.standard_normal(n)
    ctx.scan_set_model(A, w, 4)
    monkeypatch.setenv("MMG_SCAN_FUSED_LINEAR", "0")    # only the production kernel emits the linear by-products
    for hi in (2, 3):                                   # genotype alphabet {0,1} / {0,1,2}
        snps = rng.randint(0, hi, size=(m, n)).astype(np.int8)
        g = ctx.geno(snps)
        monkeypatch.delenv("MMG_SCAN_KERNEL", raising=False)
        base = ctx.scan(g, 1e6, n - 2, stats=True)
        monkeypatch.setenv("MMG_SCAN_KERNEL", variant)
        alt = ctx.scan(g, 1e6, n - 2, stats=True)
        monkeypatch.delenv("MMG_SCAN_KERNEL", raising=False)
        for k in ("den", "rss", "ps"):
            assert np.array_equal(base[k], alt[k]), (variant, hi, k)
        S = snps[:64].astype(np.float64)
        assert rel(base["den"][:64], np.einsum("ij,ij->i", S @ A, S)) < 1e-8
        g.close()


def test_scan_wide_genotype_values_use_the_64bit_epilogue(ctx):
    """|s| up to 100: the store's tracked max |s| sends the scan down the 64-bit epilogue (the 24-bit one
    would overflow); den still matches float64."""
    rng = np.random.RandomState(6)
    n, m = 700, 600
    B = rng.standard_normal((n, 10)) / 4
    A = np.eye(n) + B @ B.T / n
    ctx.scan_set_model(A, rng.standard_normal(n), 4)
    snps = rng.randint(-100, 101, size=(m, n)).astype(np.int8)
    out = ctx.scan(ctx.geno(snps), 1e12, n - 2, stats=True)
    S = snps.astype(np.float64)
    assert rel(out["den"], np.einsum("ij,ij->i", S @ A, S)) < 1e-8


def test_eigh_block_jacobi_matches_direct(ctx, monkeypatch):
    """The solver used beyond rocSOLVER's index range (block Jacobi over dsyevd pair problems), forced at a
    small size with ragged blocks: eigenvalues equal the direct solver's to 1e-12 relative to the spectral
    norm, eigenvectors are orthonormal and satisfy K v = lambda v; a kinship-like spectrum (a few large
    eigenvalues + bulk) and a clustered one."""
    rng = np.random.RandomState(12)
    n = 1111
    X = (rng.random_sample((n, 3000)) < 0.4).astype(np.float64)
    K1 = X @ X.T / 3000.0
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    K2 = (Q * np.repeat([1.0, 1.0 + 1e-9, 2.0, 5.0], [400, 300, 400, 11])) @ Q.T
    for K in (K1, 0.5 * (K2 + K2.T)):
        monkeypatch.delenv("MMG_EIGH_BLOCK", raising=False)
        v0, _ = ctx.eigh(K)
        monkeypatch.setenv("MMG_EIGH_BLOCK", "320")          # 4 blocks: 320, 320, 320, 151
        v1, U = ctx.eigh(K)
        monkeypatch.delenv("MMG_EIGH_BLOCK", raising=False)
        scale = np.abs(v0).max()
        assert np.max(np.abs(v1 - v0)) < 1e-12 * scale
        assert np.all(np.diff(v1) >= 0)
        assert np.max(np.abs(U @ U.T - np.eye(n))) < 1e-11     # rows are the eigenvectors
        assert np.max(np.abs(K @ U.T - U.T * v1)) < 1e-10 * scale


def test_background_delivery_snapshots_and_overlaps(ctx):
    """mmg_scan_deliver_begin/wait: the delivery of scan A runs on a second stream while scan B (other
    phenotype scale) executes; A's host buffers hold A's results, B's hold B's -- bit-identical to the blocking
    fetch -- with and without an RCCL communicator (world 1: the all-gather degenerates to a copy)."""
    from mixmogam_amd import dist as mdist
    rng = np.random.RandomState(3)
    n, m = 600, 5000
    B = rng.standard_normal((n, 8)) / 4
    A = np.eye(n) + B @ B.T / n
    ctx.scan_set_model(A, rng.standard_normal(n), 4)
    g = ctx.geno((rng.random_sample((m, n)) < 0.3).astype(np.int8))
    refA = ctx.scan(g, 1e5, n - 2)
    refB = ctx.scan(g, 3e5, n - 2)
    os.environ.setdefault("MASTER_PORT", "29611")
    coll = mdist.RcclCollectives(ctx, 0, 1, mdist.file_bootstrap(0, 1))
    try:
        for comm in (None, coll.h):
            bufA = [ctx.pinned_empty(m) for _ in range(3)]
            bufB = [ctx.pinned_empty(m) for _ in range(3)]
            for b in bufA + bufB:
                b[:] = -1.0
            ctx.scan(g, 1e5, n - 2, fetch=False)
            ctx.scan_deliver_begin(bufA, count=m, comm=comm)
            ctx.scan(g, 3e5, n - 2, fetch=False)              # overwrites the device result arrays
            ctx.scan_deliver_begin(bufB, count=m, comm=comm)  # waits for A's delivery, then starts B's
            ctx.scan_deliver_wait()
            for k, name in enumerate(("rss", "f_stats", "ps")):
                assert np.array_equal(bufA[k], refA[name]), (comm, name)
                assert np.array_equal(bufB[k], refB[name]), (comm, name)
    finally:
        coll.close()


def test_adaptive_digit_schedule(ctx, monkeypatch):
    """Default model (ndigits = 0): three digit planes for every SNP, the fourth only for the SNPs whose p could
    move by more than 2.5e-7 at six sigma of the 22-bit rounding noise.  Those are bit-identical to the explicit
    4-plane scan, the others agree to 1e-6 on p; the refined set validates the error model (sigma ratio < 1).  A
    target of 0 refines everything and MMG_SCAN_ADAPTIVE=0 switches the schedule off -- both bit-identical to the
    4-plane scan everywhere.  A SNP nearly collinear with a covariate (tiny den) must be among the refined."""
    rng = np.random.RandomState(21)
    n, m = 1300, 6000
    snps = struct_snps(rng, n, m)
    cof = snps[17].astype(np.float64)
    snps[18] = snps[17]
    snps[18, :3] ^= 1                                     # differs from the covariate in 3 individuals: den ~ 0
    y = rng.randn(n) + 0.4 * snps[3] + 0.3 * snps[10]
    X = np.hstack([np.ones((n, 1)), cof.reshape(n, 1)])
    est = orc.get_estimates(y, X, orc.scale_k(orc.scale_k(orc.calc_ibs_kinship(snps))))
    prep = orc.scan_prepare(y, X, est["H_sqrt_inv"])
    ref = orc.scan_closed(snps, prep)
    g = ctx.geno(snps)
    ctx.scan_set_model(prep["A"], prep["w"], 4)
    full = ctx.scan(g, prep["h0_rss"], n - 3, stats=True)
    assert not ctx.scan_last_stats()["adaptive"]
    # the digit planes on their own (round 4 adds an fp64 tier behind them: off for the bit-for-bit statements)
    monkeypatch.setenv("MMG_SCAN_EXACT", "0")
    ctx.scan_set_model(prep["A"], prep["w"], 0)
    ada = ctx.scan(g, prep["h0_rss"], n - 3, stats=True)
    st = ctx.scan_last_stats()
    assert st["adaptive"] and not st["fell_back"] and 0 < st["n_refined"] < m // 2 and st["sigma_ratio_max"] < 1.0
    assert st["n_exact"] == 0
    same = ada["den"] == full["den"]
    assert int(same.sum()) >= st["n_refined"] and same[18]
    for k in ("rss", "f_stats", "ps"):
        assert np.array_equal(ada[k][same], full[k][same]), k
    ok = ref["den"] > 1e-6 * ref["den"].max()
    assert rel(ada["ps"][ok], full["ps"][ok]) < 1e-6 and rel(ada["ps"][ok], ref["ps"][ok]) < 1e-6
    monkeypatch.setenv("MMG_SCAN_ADAPT_TARGET", "0")
    everything = ctx.scan(g, prep["h0_rss"], n - 3, stats=True)
    assert ctx.scan_last_stats()["fell_back"]
    monkeypatch.delenv("MMG_SCAN_ADAPT_TARGET")
    # ... and with the tier: the SNPs whose den four planes cannot pin down come from the fp64 matrix and sit on the
    # oracle's values; everything else is what the planes gave
    monkeypatch.delenv("MMG_SCAN_EXACT")
    ctx.scan_set_model(prep["A"], prep["w"], 0)
    tier = ctx.scan(g, prep["h0_rss"], n - 3, stats=True)
    st2 = ctx.scan_last_stats()
    moved = tier["den"] != ada["den"]
    assert 0 < st2["n_exact"] < m // 4 and 0 < int(moved.sum()) <= st2["n_exact"]
    assert np.max(np.abs(tier["den"][moved] - ref["den"][moved])) <= 1e-11 * ref["den"].max()
    assert np.max(np.abs(tier["den"][moved] - ref["den"][moved])) <= np.max(np.abs(ada["den"][moved] - ref["den"][moved]))
    assert rel(tier["ps"][ok], ref["ps"][ok]) < 1e-6
    monkeypatch.setenv("MMG_SCAN_ADAPTIVE", "0")
    ctx.scan_set_model(prep["A"], prep["w"], 0)
    off = ctx.scan(g, prep["h0_rss"], n - 3, stats=True)
    monkeypatch.delenv("MMG_SCAN_ADAPTIVE")
    for k in ("den", "rss", "f_stats", "ps"):
        assert np.array_equal(everything[k], full[k]), k
        assert np.array_equal(off[k], full[k]), k


@pytest.mark.parametrize("alphabet", [(0, 3), (-1, 2), (0, 6)])
def test_adaptive_digit_schedule_other_genotype_alphabets(ctx, alphabet):
    """The bias and sigma terms of the adaptive schedule use sum s and sum s^2 (not allele counts): 0/1/2
    genotypes, centred -1/0/1 ones and 0..5 dosages all stay within the 2.5e-7 design target of the full scan
    and validate the error model (observed / six-sigma < 1, no fallback)."""
    rng = np.random.RandomState(31)
    n, m = 1500, 5000
    lo, hi = alphabet
    snps = rng.randint(lo, hi, size=(m, n)).astype(np.int8)
    B = rng.standard_normal((n, 25)) / 5
    A = np.eye(n) * 1.5 + B @ B.T / n
    A[np.triu_indices(n, 1)] += 0.02 * rng.standard_normal(n * (n - 1) // 2)
    A = 0.5 * (A + A.T)
    w = rng.standard_normal(n)
    g = ctx.geno(snps)
    ctx.scan_set_model(A, w, 4)
    full = ctx.scan(g, 1e7, n - 2, stats=True)
    ctx.scan_set_model(A, w, 0)
    ada = ctx.scan(g, 1e7, n - 2, stats=True)
    st = ctx.scan_last_stats()
    assert st["adaptive"] and not st["fell_back"] and st["sigma_ratio_max"] < 1.0 and st["n_refined"] > 0
    ok = full["ps"] > 1e-290
    assert rel(ada["ps"][ok], full["ps"][ok]) < 2.5e-7
    S = snps[:32].astype(np.float64)
    assert rel(full["den"][:32], np.einsum("ij,ij->i", S @ A, S)) < 1e-8


def test_adaptive_schedule_is_refused_for_matrices_with_flat_blocks(ctx):
    """The error model of the adaptive schedule needs a lowest digit plane without structure.  A matrix whose
    off-diagonal is constant over whole tiles (here: everywhere) has a constant lowest digit; the per-tile mean
    check at quantisation time catches it and the default model then runs every plane for every SNP."""
    rng = np.random.RandomState(41)
    n, m = 900, 2000
    A = np.full((n, n), 0.0123456789)                      # the maximum: its lowest digit is 0 ...
    A[256:512, 0:256] = A[0:256, 256:512] = 0.00777777     # ... this block's is a constant that is not
    A[512:768, 256:512] = A[256:512, 512:768] = 0.00313131
    A += np.eye(n) * 2.0
    w = rng.standard_normal(n)
    snps = (rng.random_sample((m, n)) < 0.4).astype(np.int8)
    g = ctx.geno(snps)
    ctx.scan_set_model(A, w, 4)
    full = ctx.scan(g, 1e7, n - 2, stats=True)
    ctx.scan_set_model(A, w, 0)
    dflt = ctx.scan(g, 1e7, n - 2, stats=True)
    assert not ctx.scan_last_stats()["adaptive"]
    for k in ("den", "rss", "ps"):
        assert np.array_equal(dflt[k], full[k]), k

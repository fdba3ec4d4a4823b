"""GPU parity tests (-m gpu) added in round 2: the eigen-rotated store and the multi-phenotype scan (SURVEY 8e
row 5), replicates through Z, emma(), get_ML, _get_eigen_R_, var_perc / h0_betas, the literal-fp32 reference
(loose), genotype validation and the per-entry device binding.  Same bars as test_gpu_parity.py: bit-exact for
integers, p-values within 1e-6 relative of the double-promoted reference, tolerance written at each assert."""
import ctypes as C
import threading

import numpy as np
import pytest

from conftest import load_case, load_extras

pytestmark = pytest.mark.gpu

orc = pytest.importorskip("oracle.emmax_oracle")


@pytest.fixture(scope="module")
def ctx():
    from mixmogam_amd import _lib
    return _lib.get_context()


@pytest.fixture(scope="module")
def ex():
    return load_extras()


def rel(a, b):
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))) if len(a) else 0.0


# ------------------------------------------------------------------ eigen-rotated store
@pytest.mark.parametrize("n,m,hi", [(150, 700, 2), (300, 1000, 3), (257, 513, 2), (1100, 2500, 2), (64, 5, 3)])
def test_rot_store_equals_float64_rotation(ctx, n, m, hi):
    """T[i][m] = u_i . s_m from the int8 digit GEMM (4 unsigned 7-bit digits per eigenvector, exact integer
    accumulation) against a float64 matrix product.  Every entry of u_i is rounded to 2^-27 max|u_i| (uniform, +-half
    a step), so the error of T is a sum of sum(s^2) such roundings: sigma = 2^-28 max|u| sqrt(sum s^2 / 3); asserted
    at 8 sigma (~1e-8 of |T|; round 2's balanced base-256 digits carried three more bits at 6-9 % more time)."""
    rng = np.random.RandomState(n + m)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    V = np.ascontiguousarray(Q.T)                           # rows orthonormal, like eigh's eigenvectors
    snps = rng.randint(0, hi, size=(m, n)).astype(np.int8)
    g = ctx.geno(snps)
    rot = ctx.rot(V, m).load(g)
    T = rot.fetch()
    ref = V @ snps.T.astype(np.float64)
    assert T.shape == (n, m)
    tol = 8 * 2.0 ** -28 * np.max(np.abs(V)) * np.sqrt(np.max((snps.astype(np.float64) ** 2).sum(1)) / 3)
    assert tol < 4e-8 * np.max(np.abs(ref))
    assert np.max(np.abs(T - ref)) < tol
    assert np.max(np.abs(rot.fetch(3, 2) - ref[:, 3:5])) < tol
    # reloading another (smaller) block reuses the store
    g2 = ctx.geno(snps[: m // 2 + 1])
    rot.load(g2)
    assert np.max(np.abs(rot.fetch() - ref[:, : m // 2 + 1])) < tol
    rot.close(); g.close(); g2.close()


@pytest.mark.parametrize("P,q", [(1, 1), (3, 1), (8, 2), (11, 3), (5, 4), (16, 1), (17, 2), (41, 1), (9, 4), (16, 3)])
def test_scan_multi_c_abi_vs_numpy(ctx, P, q):
    """mmg_emmax_scan_multi on arbitrary coefficient vectors against the same sums in numpy float64 (the sums are
    plain fp64 FMAs over N terms: 1e-11 relative on den/dot, so 1e-9 on rss / F away from cancellation)."""
    # batches: 16 per pass on the fp64 matrix pipe for q <= 2, 8 beyond; 41 phenotypes = three batches, i.e. the
    # double-buffered result sets are reused; n = 333 leaves a partial group of coordinates, n = 64 has no zero rows
    # behind the last eigenvector
    rng = np.random.RandomState(10 * P + q)
    n, m = (333, 1500) if P != 9 else (64, 300)
    Qm, _ = np.linalg.qr(rng.standard_normal((n, n)))
    V = np.ascontiguousarray(Qm.T)
    snps = rng.randint(0, 2, size=(m, n)).astype(np.int8)
    snps[5] = 1                                             # monomorphic row
    g = ctx.geno(snps)
    rot = ctx.rot(V, m).load(g)
    d = rng.uniform(0.2, 2.0, size=(P, n))
    omega = rng.standard_normal((P, n)) * 0.05
    G = rng.standard_normal((P, q, n)) * 0.02
    h0 = rng.uniform(50, 100, size=P)
    out = ctx.scan_multi(rot, d, omega, G, h0, n - q - 1)
    T = V @ snps.T.astype(np.float64)
    aq = d @ (T * T)
    den = aq - sum((G[:, c, :] @ T) ** 2 for c in range(q))
    dot = omega @ T
    rss = h0[:, None] - dot * dot / den
    F = (h0[:, None] / rss - 1) * (n - q - 1)
    assert out["ps"].shape == (P, m)
    assert rel(out["rss"], rss) < 1e-10
    # F = (h0/rss - 1) nu cancels for small F (dot is a sum of N signed terms; the rotated store carries ~1e-9 of
    # |tau| per entry): absolute 1e-7 below F = 1, relative above; p is insensitive there and gets the 1e-7 bar
    assert np.max(np.abs(out["f_stats"] - F) / np.maximum(F, 1.0)) < 1e-7
    assert rel(out["ps"], orc.f_sf(np.maximum(F, 0), 1, n - q - 1)) < 1e-7
    rot.close(); g.close()


def test_emmax_multi_vs_reference_loop(ctx, ex):
    """The product's multi-phenotype scan against the REFERENCE's loop of emmax() runs (tests/golden/extras_n150,
    six phenotypes with different variance ratios; without and with a cofactor): p-values 1e-6 relative."""
    from mixmogam_amd import linear_models as lm
    for tag, cof in (("multi", None), ("multic", [ex["multi_cof"]])):
        res = lm.emmax_multi(list(ex["snps"]), ex["multi_ys"], ex["ibs_scaled"], cofactors=cof, ctx=ctx)
        assert rel(res["ps"], ex["dbl_%s_ps" % tag]) < 1e-6
        assert rel(res["rss"], ex["dbl_%s_rss" % tag]) < 1e-7
        assert np.max(np.abs(res["var_perc"] - ex["dbl_%s_var_perc" % tag])) < 1e-8
        for k in ("h0_rss", "pseudo_heritability", "max_ll"):
            assert rel(res[k], ex["dbl_%s_%s" % (tag, k)]) < 1e-7, k
        # ... and against the product's own single-phenotype path (quadratic-form GEMM), phenotype by phenotype
        one = lm.emmax(list(ex["snps"]), list(ex["multi_ys"][3]), ex["ibs_scaled"], cofactors=cof, ctx=ctx)
        assert rel(res["ps"][3], one["ps"]) < 1e-6
        # chunked over the SNP axis (small store budget) and from a device-resident store: identical bits
        g = ctx.geno(ex["snps"])
        res2 = lm.emmax_multi(g, ex["multi_ys"], ex["ibs_scaled"], cofactors=cof, ctx=ctx, max_store_bytes=256 * 8 * 192)
        g.close()
        assert np.array_equal(res2["ps"], res["ps"])


def test_emmax_multi_random_sizes_vs_oracle(ctx):
    rng = np.random.RandomState(77)
    n, m, P = 420, 1800, 10
    pops = rng.randint(0, 3, size=n)
    freqs = rng.uniform(0.1, 0.9, size=(m, 3))
    snps = (rng.random_sample((m, n)) < freqs[:, pops]).astype(np.int8)
    snps = snps[(snps.sum(1) > 0) & (snps.sum(1) < n)]
    from mixmogam_amd import kinship, linear_models as lm
    K = kinship.calc_ibs_kinship(snps, ctx=ctx)
    ys = np.asarray([snps[rng.choice(len(snps), 4)].astype(float).sum(0) * rng.uniform(0.2, 2) + rng.randn(n)
                     for _ in range(P)])
    snps[17] = 1                                            # monomorphic: collinear with the intercept
    res = lm.emmax_multi(snps, ys, K, ctx=ctx)
    ref = orc.emmax_multi(snps, ys, K)
    assert np.all(res["ps"][:, 17] == 1.0) and np.all(res["rss"][:, 17] == res["h0_rss"])   # :1308,1329
    keep = np.arange(len(snps)) != 17
    assert rel(res["ps"][:, keep], ref["ps"][:, keep]) < 1e-6
    assert rel(res["delta"], ref["delta"]) < 1e-6


# ------------------------------------------------------------------ small parity holes of round 1
def test_eigen_R_values_vs_golden(ctx, case):
    """_get_eigen_R_ (linear_models.py:600-615) on the device against the reference's values."""
    from mixmogam_amd import linear_models as lm
    lmm = lm.LinearMixedModel(list(case["y"]), ctx=ctx)
    lmm.add_random_effect(case["dbl_ibs_scaled"])
    if case["cof"] is not None:
        for c in case["cof"]:
            lmm.add_factor(c)
    eR = lmm._get_eigen_R_(X=lmm.X)
    assert eR["values"].shape == case["dbl_eig_R_values"].shape
    assert np.all(np.diff(eR["values"]) >= -1e-12)                       # ascending, as scipy's eigh returns them
    assert np.max(np.abs(eR["values"] - case["dbl_eig_R_values"])) < 1e-9
    assert eR["vectors"].shape == (lmm.n - lmm.X.shape[1], lmm.n)
    # rows orthonormal and orthogonal to X (S annihilates X)
    assert np.max(np.abs(eR["vectors"] @ eR["vectors"].T - np.eye(len(eR["values"])))) < 1e-9
    assert np.max(np.abs(eR["vectors"] @ lmm.X)) < 1e-8 * np.max(np.abs(lmm.X))
    # the explicit eig_R route of get_estimates gives the golden REML scalars too
    est = lmm.get_estimates(lmm._get_eigen_L_(), method="REML", eig_R=eR, ngrids=100, use_eig_R=True)
    assert rel(est["delta"], case["dbl_reml_delta"]) < 1e-7


def test_emmax_var_perc_h0_betas_and_literal_reference(ctx, case):
    from mixmogam_amd import linear_models as lm
    res = lm.emmax(list(case["snps"]), list(case["y"]), case["dbl_ibs_scaled"], cofactors=case["cof"], ctx=ctx)
    assert np.max(np.abs(res["var_perc"] - case["dbl_emmax_var_perc"])) < 1e-8
    assert rel(res["h0_betas"], case["dbl_emmax_h0_betas"]) < 1e-6
    Fg = case["dbl_emmax_f_stats"]
    assert np.max(np.abs(res["f_stats"] - Fg) / np.maximum(Fg, 1.0)) < 1e-6   # relative above F = 1, absolute below
    # the reference AS WRITTEN ('single' storage): its own p-values are off from its float64 evaluation by up to
    # 3.8e-2 relative (measured on these cases), so the HIP path is compared with it loosely: 5e-2 on p >= 1e-12,
    # and the ranking of the ten smallest p-values agrees
    lit = case["lit_emmax_ps"]
    ok = lit > 1e-12
    assert rel(res["ps"][ok], lit[ok]) < 5e-2
    assert set(np.argsort(res["ps"])[:5]) <= set(np.argsort(lit)[:10])


def test_emmax_with_replicates_Z_vs_golden(ctx, ex):
    """emmax(Z=...) (linear_models.py:1796,1296-1297): 190 measurements of 150 accessions."""
    from mixmogam_amd import linear_models as lm
    res = lm.emmax(list(ex["snps"]), list(ex["z_y"]), ex["ibs_scaled"], Z=ex["z_Z"], ctx=ctx)
    assert rel(res["ps"], ex["dbl_z_ps"]) < 1e-6
    assert rel(res["h0_rss"], ex["dbl_z_h0_rss"]) < 1e-8
    assert rel(res["pseudo_heritability"], ex["dbl_z_pseudo_heritability"]) < 1e-7
    assert np.max(np.abs(res["var_perc"] - ex["dbl_z_var_perc"])) < 1e-8


def test_replicates_through_the_containers(ctx, ex):
    """coordinate_w_phenotype_data keeps replicated measurements (snpsdata.py:2236-2240) and
    get_incidence_matrix turns them into Z: the container route gives the golden Z run."""
    from mixmogam_amd import linear_models as lm, phenotypeData as pd, snpsdata
    n = 150
    accs = ["a%03d" % i for i in range(n)]
    reps = np.argmax(ex["z_Z"], axis=1)
    order = np.random.RandomState(1).permutation(len(reps))            # phenotype file in arbitrary order
    phend = pd.phenotype_data({1: {"name": "t", "ecotypes": [accs[reps[i]] for i in order],
                                   "values": [float(ex["z_y"][i]) for i in order]}})
    sd = snpsdata.construct_snps_data_set(ex["snps"], list(range(len(ex["snps"]))), [1] * len(ex["snps"]), accs)
    sd.coordinate_w_phenotype_data(phend, 1)
    assert len(phend.get_values(1)) == len(reps)
    Z = phend.get_incidence_matrix(1)
    assert np.array_equal(Z, ex["z_Z"])
    # replicates of one accession may come out in a different order than the fixture's: compare as multisets per
    # accession, then run with the fixture's order
    got = sorted(zip(phend.get_ecotypes(1), phend.get_values(1)))
    want = sorted(zip([accs[r] for r in reps], [float(v) for v in ex["z_y"]]))
    assert got == want


def test_emma_and_ml_vs_golden(ctx, ex):
    from mixmogam_amd import linear_models as lm
    y = list(ex["multi_ys"][2])
    res = lm.emma(ex["snps"][:12], y, ex["ibs_scaled"], ctx=ctx)
    for k in ("ps", "f_stats", "vgs", "ves", "var_perc", "max_lls", "rss"):
        assert rel(res[k], ex["dbl_emma_" + k]) < 2e-6, k
    assert np.max(np.abs(np.asarray(res["betas"]) - ex["dbl_emma_betas"])) < 1e-6
    lmm = lm.LinearMixedModel(y, ctx=ctx)
    lmm.add_random_effect(ex["ibs_scaled"])
    ml = lmm.get_ML()
    for k in ("max_ll", "delta", "ve", "vg", "pseudo_heritability"):
        assert rel(ml[k], ex["dbl_ml_" + k]) < 1e-6, k
    assert rel(ml["beta"], ex["dbl_ml_beta"]) < 1e-6
    rl = lmm.get_REML()
    assert rel(rl["delta"], ex["dbl_reml100_delta"]) < 1e-7 and rel(rl["max_ll"], ex["dbl_reml100_max_ll"]) < 1e-8


# ------------------------------------------------------------------ ingest validation, device binding
def test_non_integral_or_out_of_range_genotypes_are_rejected(ctx):
    from mixmogam_amd import _lib
    snps = np.random.RandomState(0).randint(0, 3, size=(40, 70)).astype(np.float64)
    ctx.geno(snps).close()                                            # integral floats are fine
    for bad in (0.5, 200.0, np.nan, -128.0):
        s2 = snps.copy()
        s2[7, 3] = bad
        for dt in (np.float32, np.float64):
            with pytest.raises(_lib.MixmogamHipError, match="integers in"):
                ctx.geno(s2.astype(dt))
    with pytest.raises(ValueError):
        ctx.geno(np.full((3, 70), 200, dtype=np.int16))               # would wrap to -56 in int8
    with pytest.raises(_lib.MixmogamHipError):                        # found on the device; the rows are rolled back
        ctx.geno(np.full((3, 70), -128, dtype=np.int8))
    assert np.array_equal(ctx.geno(np.full((3, 70), 2, dtype=np.int64)).download(), np.full((3, 70), 2, dtype=np.int8))


def test_entry_points_bind_their_device_from_any_thread(ctx):
    """A helper thread (HIP's current device is per thread) uploads and scans through the same library: the
    allocations land on the context's device and the results equal the main thread's."""
    rng = np.random.RandomState(3)
    n, m = 200, 900
    snps = rng.randint(0, 2, size=(m, n)).astype(np.int8)
    B = rng.standard_normal((n, 5))
    A = np.eye(n) + B @ B.T / n
    w = rng.standard_normal(n)
    ctx.scan_set_model(A, w, 4)
    g = ctx.geno(snps)
    base = ctx.scan(g, 1e5, n - 2)["ps"]
    g.close()
    box = {}

    def work():
        try:
            g2 = ctx.geno(snps)                                       # hipMalloc + copies from a fresh thread
            box["ps"] = ctx.scan(g2, 1e5, n - 2)["ps"]
            box["counts"] = ctx.kinship_ibs_counts(g2)
            g2.close()
        except Exception as e:                                        # pragma: no cover
            box["err"] = e

    t = threading.Thread(target=work)
    t.start(); t.join()
    assert "err" not in box, box.get("err")
    assert np.array_equal(box["ps"], base)
    assert np.array_equal(box["counts"], orc.ibs_counts(snps))


def test_structured_generator_matches_oracle_and_adaptive_scan_on_structured_data(ctx):
    """mmg_geno_fill_structured == its host twin bit for bit; and the adaptive digit schedule on structured genotypes
    with many strong hits at N = 5000 ('the regime real GWAS lives in'): the top hits and a random
    sample are within 1e-6 of the float64 evaluation of the reference's per-SNP arithmetic, whatever the schedule
    chose to refine, and the scan equals the all-planes scan to 1e-6 everywhere."""
    from mixmogam_amd import kinship, linear_models as lm
    g0 = ctx.geno(M=700, N=333)
    g0.fill_structured(9, m_global0=123, npop=3, spread_q16=9830)
    assert np.array_equal(g0.download(), orc.hash_genotypes_structured(123, 823, 333, 9, 3, 9830))
    g0.fill_structured(9, m_global0=0, npop=5, spread_q16=20000)
    assert np.array_equal(g0.download(), orc.hash_genotypes_structured(0, 700, 333, 9, 5, 20000))
    g0.close()
    n, m = 5000, 120000
    g = ctx.geno(M=m, N=n).fill_structured(31, 0, npop=3, spread_q16=9830)
    counts = ctx.kinship_ibs_counts(g)
    K = kinship.scale_k(counts.astype(np.float64) / (2.0 * m) + 0.5)
    rng = np.random.RandomState(32)
    causal = np.sort(rng.choice(m, 110, replace=False))
    rows = orc.hash_genotypes_structured(0, m, n, 31)[causal].astype(np.float64) if False else \
        np.vstack([orc.hash_genotypes_structured(int(c), int(c) + 1, n, 31) for c in causal]).astype(np.float64)
    assert np.array_equal(rows.astype(np.int8), g.download_rows(causal))
    gen = rng.exponential(1.0, size=110) @ (rows - rows.mean(1, keepdims=True))
    err = rng.standard_normal(n)
    y = gen + err * np.sqrt(0.25 * gen.var(ddof=1) / err.var(ddof=1))
    y = (y - y.mean()) / y.std()
    lmm = lm.LinearMixedModel(y, ctx=ctx)
    lmm.add_random_effect(K)
    eig_L = lmm._get_eigen_L_()
    est = lmm.get_estimates(eig_L, method="REML")
    assert 0.02 < est["pseudo_heritability"] < 0.999                    # interior optimum on structured data
    prep = lmm.scan_prepare(est["H_sqrt_inv"])
    ctx.scan_set_model(prep["A"], prep["w"], 0)
    ada = ctx.scan(g, prep["h0_rss"], prep["n_p"])
    st = ctx.scan_last_stats()
    assert st["adaptive"] and (st["fell_back"] or st["n_refined"] > 0)
    ctx.scan_set_model(prep["A"], prep["w"], 4)
    allp = ctx.scan(g, prep["h0_rss"], prep["n_p"])
    ok = allp["ps"] > 1e-290
    assert int((allp["ps"] < 1e-8).sum()) >= 20 and int((allp["ps"] < 1e-5).sum()) >= 30   # many strong hits
    assert rel(ada["ps"][ok], allp["ps"][ok]) < 1e-6
    H = np.asarray(est["H_sqrt_inv"])
    Q, _ = np.linalg.qr(H @ lmm.X)
    hits = np.argsort(allp["ps"])
    hits = hits[allp["ps"][hits] > 1e-280][:12]
    sample = np.unique(np.r_[hits, rng.choice(m, 12, replace=False)])
    worst = 0.0
    for gi in sample:
        s = orc.hash_genotypes_structured(int(gi), int(gi) + 1, n, 31)[0].astype(np.float64)
        t = H @ s
        t = t - Q @ (Q.T @ t)
        rss = prep["h0_rss"] - float(t @ prep["r"]) ** 2 / float(t @ t)
        F = (prep["h0_rss"] / rss - 1) * prep["n_p"]
        p = float(orc.f_sf(np.array([F]), 1, prep["n_p"])[0])
        worst = max(worst, abs(ada["ps"][gi] / p - 1))
    assert worst < 1e-6, worst
    g.close()


def test_perm_after_scan_equals_the_standalone_test(ctx, case):
    """mmg_emmax_perm_after_scan (t.t rebuilt from the quadratic forms of the scan that just ran over the same store
    with the same H, plus 1 + q dot products per SNP) against the standalone permutation test and, where the golden
    case has one, against the reference's run: min_rss within 1e-7 relative of the exact path (den carries the
    adaptive scan's precision), max F / min p within 1e-6 of the reference."""
    from mixmogam_amd import _lib, linear_models as lm
    lmm = lm.LinearMixedModel(list(case["y"]), ctx=ctx)
    lmm.add_random_effect(case["dbl_ibs_scaled"])
    if case["cof"] is not None:
        for c in case["cof"]:
            lmm.add_factor(c)
    eig_L = lmm._get_eigen_L_()
    est = lmm.get_estimates(eig_L, method="REML")
    H = np.asarray(case["dbl_perm_H"]) if "dbl_perm_H" in case else est["H_sqrt_inv"]
    prep = lmm.scan_prepare(H)
    n = lmm.n
    idx = case["dbl_perm_idx"] if "dbl_perm_idx" in case else \
        np.array([np.random.RandomState(p).permutation(n) for p in range(17)])
    lmm_p = lm.LinearMixedModel(list(case["y"]), ctx=ctx)
    lmm_p.X, lmm_p.p = lmm.X, lmm.p
    pp = lmm_p.perm_prepare(H, perm_idx=idx)
    g = ctx.geno(case["snps"])
    with pytest.raises(_lib.MixmogamHipError):                       # no scan over this store yet
        ctx.perm(ctx.geno(case["snps"][:10]), pp["H"], pp["Ys"], pp["h0_rss"], after_scan_HtQ=prep["HtQ"])
    exact = ctx.perm(g, pp["H"], pp["Ys"], pp["h0_rss"])
    ctx.scan_set_model(prep["A"], prep["w"], 0)
    ctx.scan(g, prep["h0_rss"], prep["n_p"], fetch=False)
    fast = ctx.perm(g, pp["H"], pp["Ys"], pp["h0_rss"], after_scan_HtQ=prep["HtQ"])
    assert rel(fast, exact) < 1e-7
    if "dbl_perm_max_f_stats" in case and case["cof"] is None:
        max_f = (pp["h0_rss"] / fast - 1.0) * pp["n_p"]
        assert rel(max_f, case["dbl_perm_max_f_stats"]) < 1e-6
        assert rel(ctx.f_sf(max_f, pp["n_p"]), case["dbl_perm_min_ps"]) < 1e-6
    g.upload(case["snps"][:1], 0)                                     # any write invalidates the scan's state
    with pytest.raises(_lib.MixmogamHipError):
        ctx.perm(g, pp["H"], pp["Ys"], pp["h0_rss"], after_scan_HtQ=prep["HtQ"])
    g.close()


def test_eigen_free_reml_on_the_device_vs_golden(ctx, case):
    """mmg_reml_sums / mmg_reml_scan_model (Cholesky per delta, recursive triangular inverse, P and Py built in HBM):
    the likelihood sums equal the eig_L route's to 1e-9, the REML scalars and the p-values equal the reference's."""
    from mixmogam_amd import linear_models as lm
    lmm = lm.LinearMixedModel(list(case["y"]), ctx=ctx)
    lmm.add_random_effect(case["dbl_ibs_scaled"])
    if case["cof"] is not None:
        for c in case["cof"]:
            lmm.add_factor(c)
    eig_L = lmm._get_eigen_L_()
    reml = ctx.reml(lmm.random_effects[1][1], lmm.X, lmm.Y.reshape(-1))
    deltas = np.exp(np.array([-10.0, -3.0, 0.0, 0.7, 4.0, 10.0]))
    a = lm._SpectralSumsChol(reml).at(deltas)
    b = lm._SpectralSumsL(eig_L, lmm.X, lmm.Y.reshape(-1)).at(deltas)
    for k in range(4):
        assert rel(a[k], b[k]) < 1e-8, k
    reml.close()
    res = lmm.get_estimates_eigen_free()
    for k in ("pseudo_heritability", "ve", "vg", "max_ll"):
        assert rel(res[k], case["dbl_emmax_" + k]) < 1e-7, k
    prep = lmm.scan_model_eigen_free(res)
    res.pop("reml").close()
    assert rel(prep["h0_rss"], case["dbl_emmax_h0_rss"]) < 1e-7
    assert rel(prep["h0_betas"], case["dbl_emmax_h0_betas"]) < 1e-6
    g = ctx.geno(case["snps"])
    out = ctx.scan(g, prep["h0_rss"], prep["n_p"])
    g.close()
    assert rel(out["ps"], case["dbl_emmax_ps"]) < 1e-6


def test_eigen_free_recursive_inverse_beyond_one_block(ctx):
    """N = 9,000 > the 4,096-row base block of the recursive triangular inverse: tr H^-1 against the eigenvalues."""
    from mixmogam_amd import linear_models as lm
    rng = np.random.RandomState(4)
    n = 9000
    B = rng.standard_normal((n, 40))
    K = B @ B.T / 40 + 0.05 * np.eye(n)
    y = rng.standard_normal(n)
    X = np.ones((n, 1))
    reml = ctx.reml(K, X, y)
    deltas = np.array([0.3, 5.0])
    s1, s2, s3, s4 = lm._SpectralSumsChol(reml).at(deltas)
    reml.close()
    vals, vecs = ctx.eigh(K)
    ref = lm._SpectralSumsL({"values": vals, "vectors": vecs}, X, y).at(deltas)
    for got, want in zip((s1, s2, s3, s4), ref):
        assert rel(got, want) < 1e-8


@pytest.mark.parametrize("n", [300, 4500])
def test_eigen_free_reports_an_indefinite_matrix(ctx, n):
    """K + delta I not positive definite (a kinship with a negative eigenvalue and a tiny delta): both Cholesky paths
    (rocSOLVER below 4,096 rows, the blocked one above) fail with an error instead of returning numbers."""
    from mixmogam_amd import _lib
    rng = np.random.RandomState(n)
    B = rng.standard_normal((n, 8))
    K = B @ B.T / 8 - 0.5 * np.eye(n)                       # eigenvalues down to -0.5
    reml = ctx.reml(K, np.ones((n, 1)), rng.standard_normal(n))
    with pytest.raises(_lib.MixmogamHipError, match="positive definite"):
        reml.sums([1e-3])
    s = reml.sums([1.0])                                     # K + I is fine again
    assert np.all(np.isfinite([v[0] for v in s[:4]]))
    reml.close()


# ------------------------------------------------------------------ exact GRM on the int8 matrix cores
def _grm_f64(snps):
    s = snps.astype(np.float64)
    z = (s - s.mean(1, keepdims=True)) / s.std(1)[:, None]
    return z.T @ z


def test_grm_exact_route_vs_golden(ctx, case):
    """calc_ibd_kinship through mmg_kin_acc_add_grm (weight digits folded into one int8 operand, 4 exact GEMMs):
    the double-promoted reference's GRM to 3e-9 -- the weights carry 31 bits relative to the LARGEST 1/std^2 (the
    rarest SNP of these structured cases, 40x the typical weight) -- where the fp32-MFMA kernel was good to 2e-5."""
    from mixmogam_amd import kinship
    k = kinship.calc_ibd_kinship(case["snps"], ctx=ctx)
    assert np.max(np.abs(k - case["dbl_ibd_scaled"])) < 3e-9


@pytest.mark.parametrize("hi,n,m", [(2, 300, 2000), (3, 257, 1500), (5, 130, 900), (10, 100, 500)])
def test_grm_exact_route_alphabets_and_chunks(ctx, monkeypatch, hi, n, m):
    """0/1 (8-bit digits), 0/1/2 (7-bit), 0..4 (6-bit) genotypes against float64; wider alphabets fall back to the
    fp32-MFMA kernel (1e-5); several passes over the SNP axis (MMG_KIN_CHUNK) and accumulation over two stores give
    the same matrix."""
    rng = np.random.RandomState(hi * 1000 + n)
    snps = rng.randint(0, hi, size=(m, n)).astype(np.int8)
    snps = snps[snps.std(1) > 0]
    ref = _grm_f64(snps)
    tol = 1e-9 if hi <= 5 else 2e-5
    g = ctx.geno(snps)
    acc = ctx.kinship_accumulator(n)
    acc.add_grm(g)
    k1, cnt = acc.fetch()
    assert cnt == len(snps)
    assert np.max(np.abs(k1 - ref)) < tol * np.max(np.abs(ref))
    acc.close()
    monkeypatch.setenv("MMG_KIN_CHUNK", "256")
    acc = ctx.kinship_accumulator(n)
    half = len(snps) // 2
    ga, gb = ctx.geno(snps[:half]), ctx.geno(snps[half:])
    acc.add_grm(ga); acc.add_grm(gb)
    k2, cnt = acc.fetch()
    monkeypatch.delenv("MMG_KIN_CHUNK")
    assert cnt == len(snps)
    assert np.max(np.abs(k2 - ref)) < tol * np.max(np.abs(ref))
    for x in (g, ga, gb):
        x.close()
    acc.close()


def test_grm_monomorphic_snp_is_an_error(ctx):
    from mixmogam_amd import _lib, kinship
    snps = np.random.RandomState(1).randint(0, 2, size=(50, 40)).astype(np.int8)
    snps[7] = 1
    with pytest.raises(_lib.MixmogamHipError, match="std == 0"):
        kinship.calc_ibd_kinship(snps, ctx=ctx)


def test_emmax_takes_the_eigen_free_route_above_the_threshold(ctx, case, monkeypatch):
    """lm.emmax() switches to the Cholesky route by N alone (threshold lowered for the test): same p-values."""
    from mixmogam_amd import linear_models as lm
    monkeypatch.setattr(lm, "EIGEN_FREE_MIN_N", 100)
    res = lm.emmax(list(case["snps"]), list(case["y"]), case["dbl_ibs_scaled"], cofactors=case["cof"], ctx=ctx)
    assert res["timings"]["eig_L"] == 0.0
    assert rel(res["ps"], case["dbl_emmax_ps"]) < 1e-6
    for k in ("pseudo_heritability", "ve", "vg", "max_ll"):
        assert rel(res[k], case["dbl_emmax_" + k]) < 1e-7, k


def test_indefinite_kinship_falls_back_to_the_eigen_route(ctx):
    """A user-supplied kinship with negative eigenvalues: the device-built scan model (Cholesky of K + delta I) is not
    available, emmax() still answers through H_sqrt_inv -- same p-values as with the device model switched off."""
    import warnings
    from mixmogam_amd import linear_models as lm
    rng = np.random.RandomState(8)
    n, m = 200, 400
    snps = rng.randint(0, 2, size=(m, n)).astype(np.int8)
    snps = snps[snps.std(1) > 0]
    B = rng.standard_normal((n, 30))
    K = B @ B.T / 30
    K -= 0.3 * np.outer(B[:, 0], B[:, 0]) / 30 * 4          # push one direction negative
    K = 0.5 * (K + K.T)
    assert np.linalg.eigvalsh(K).min() < -1e-3
    y = rng.standard_normal(n)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = lm.emmax(snps, y, K, ctx=ctx)
        lm.DEVICE_SCAN_MODEL = False
        try:
            b = lm.emmax(snps, y, K, ctx=ctx)
        finally:
            lm.DEVICE_SCAN_MODEL = True
    ok = np.isfinite(b["ps"])
    assert np.array_equal(np.isfinite(a["ps"]), ok)
    assert rel(a["ps"][ok], b["ps"][ok]) < 1e-6


def test_perm_plan_equals_the_one_shot_test(ctx):
    """mmg_perm_plan_*: the SNP-independent half prepared once, run over several stores -- the same bits as the one-shot
    entry points, stand-alone and after a scan."""
    from mixmogam_amd import linear_models as lm
    case = load_case("struct_n300_s2")
    lmm = lm.LinearMixedModel(list(case["y"]), ctx=ctx)
    lmm.add_random_effect(case["dbl_ibs_scaled"])
    H = np.asarray(case["dbl_perm_H"])
    prep = lmm.scan_prepare(H)
    lmm_p = lm.LinearMixedModel(list(case["y"]), ctx=ctx)
    pp = lmm_p.perm_prepare(H, perm_idx=case["dbl_perm_idx"])
    plan = ctx.perm_plan(pp["H"], pp["Ys"], pp["h0_rss"])
    ga, gb = ctx.geno(case["snps"][:1700]), ctx.geno(case["snps"][1700:])
    for g in (ga, gb):
        assert np.array_equal(plan.run(g), ctx.perm(g, pp["H"], pp["Ys"], pp["h0_rss"]))
    ctx.scan_set_model(prep["A"], prep["w"], 0)
    for g in (ga, gb):
        ctx.scan(g, prep["h0_rss"], prep["n_p"], fetch=False)
        a = plan.run(g, after_scan_HtQ=prep["HtQ"])
        ctx.scan(g, prep["h0_rss"], prep["n_p"], fetch=False)
        assert np.array_equal(a, ctx.perm(g, pp["H"], pp["Ys"], pp["h0_rss"], after_scan_HtQ=prep["HtQ"]))
    both = np.minimum(plan.run(ga), plan.run(gb))
    gall = ctx.geno(case["snps"])
    assert np.array_equal(both, plan.run(gall))
    max_f = (pp["h0_rss"] / both - 1.0) * pp["n_p"]
    assert rel(max_f, case["dbl_perm_max_f_stats"]) < 1e-6
    plan.close()
    for g in (ga, gb, gall):
        g.close()


def test_run_emmax_multi_streamed_vs_per_phenotype(ctx, tmp_path):
    """hdf5_data.run_emmax_multi over an on-disk container (chunks rotated into the eigenbasis and scanned for all
    phenotypes in one pass, double-buffered ingest) == hdf5_data.run_emmax once per phenotype (the reference's shape)."""
    from mixmogam_amd import hdf5_data, simulations
    path = simulations.write_synthetic_container(str(tmp_path / "in.mmg"), 300, 3000, chunk_rows=500, num_chroms=3,
                                                 num_causals=10)
    src = hdf5_data.open_hdf5(path)
    rng = np.random.RandomState(5)
    raw1 = np.asarray(src["genot_data"]["chrom_2"]["raw_snps"][...])
    ys = np.vstack([src["phenotypes"], rng.randn(300) + 1.5 * raw1[11], rng.randn(300) + raw1[5] - raw1[70]])
    out = hdf5_data.run_emmax_multi(path, str(tmp_path / "multi.mmg"), phenotypes=ys, min_maf=0.1, chunk_size=700, ctx=ctx)
    assert len(set(np.round(out["delta"], 9))) == 3
    for p in range(3):
        one = hdf5_data.run_emmax(path, None, phenotypes=ys[p], min_maf=0.1, chunk_size=700, k=out["kinship"], ctx=ctx)
        assert rel(out["pseudo_heritability"][p], one["pseudo_heritability"]) < 1e-8
        for c in one["chrom_results"]:
            assert rel(out["chrom_results"][c]["ps"][p], one["chrom_results"][c]["ps"]) < 1e-6


def test_scan_linear_terms_from_the_gemm_equal_the_finalize_pass(ctx, monkeypatch):
    """Binary stores with 16 free padding rows take s.w, sum A_ii s_i and sum s_i from digit rows riding in the
    quadratic-form GEMM (k_scan_w4s.hip LIN) instead of a second sweep over the store: the quadratic form is the same
    integer, the linear terms are exact sums of 52-bit digit images -- p-values equal to 1e-11, s.w to 1e-13 of
    sum |s_k w_k|.  Stores with negative values, 0/1/2 genotypes or N a multiple of 256 keep the finalize pass."""
    rng = np.random.RandomState(77)
    for n, m, adaptive in [(300, 3000, 0), (1100, 2600, 4), (150, 700, 0)]:
        snps = (rng.random_sample((m, n)) < rng.uniform(0.05, 0.95, size=(m, 1))).astype(np.int8)
        snps[3] = 0
        snps[4] = 1
        B = rng.standard_normal((n, 20)) / 4
        A = np.eye(n) * rng.uniform(0.5, 3.0, size=n) + B @ B.T / n
        A = 0.5 * (A + A.T)
        w = rng.standard_normal(n) * np.exp(rng.uniform(-6, 2, size=n))      # wide dynamic range
        g = ctx.geno(snps)
        ctx.scan_set_model(A, w, adaptive)
        fused = ctx.scan(g, 5e6, n - 2, stats=True)
        monkeypatch.setenv("MMG_SCAN_FUSED_LINEAR", "0")
        plain = ctx.scan(g, 5e6, n - 2, stats=True)
        monkeypatch.delenv("MMG_SCAN_FUSED_LINEAR")
        S = snps.astype(np.float64)
        scale = np.abs(S) @ np.abs(w)
        assert np.max(np.abs(fused["dot"] - plain["dot"]) / np.maximum(scale, 1e-300)) < 1e-13
        assert np.max(np.abs(fused["dot"] - S @ w) / np.maximum(scale, 1e-300)) < 1e-13
        assert rel(fused["den"], plain["den"]) < 1e-13
        ok = plain["ps"] > 1e-280
        assert rel(fused["ps"][ok], plain["ps"][ok]) < 1e-11
        assert np.array_equal(fused["rss"][3:5], plain["rss"][3:5])          # monomorphic rows: rss = h0_rss either way
        g.close()
    # a store with a negative value must not take the shortcut (s^2 != s): results identical with and without the switch
    snps = rng.randint(-1, 2, size=(500, 300)).astype(np.int8)
    g = ctx.geno(snps)
    ctx.scan_set_model(A[:300, :300] if A.shape[0] >= 300 else np.eye(300), rng.standard_normal(300), 4)
    a = ctx.scan(g, 5e6, 298, stats=True)
    monkeypatch.setenv("MMG_SCAN_FUSED_LINEAR", "0")
    b = ctx.scan(g, 5e6, 298, stats=True)
    monkeypatch.delenv("MMG_SCAN_FUSED_LINEAR")
    for k in ("den", "dot", "ps"):
        assert np.array_equal(a[k], b[k]), k
    g.close()


@pytest.mark.parametrize("n", [240, 241, 255, 256, 257, 512, 1008, 1009])
def test_scan_at_the_edges_of_the_linear_row_condition(ctx, n):
    """16 free padding rows (N mod 256 <= 240, N not a multiple of 256) decide whether the linear terms ride in the GEMM;
    both sides of every edge against float64 numpy: den, s.w, sum s, p."""
    rng = np.random.RandomState(n)
    m = 700
    snps = (rng.random_sample((m, n)) < 0.4).astype(np.int8)
    B = rng.standard_normal((n, 12)) / 3
    A = np.eye(n) * 1.3 + B @ B.T / n
    A = 0.5 * (A + A.T)
    w = rng.standard_normal(n)
    g = ctx.geno(snps)
    ctx.scan_set_model(A, w, 4)
    out = ctx.scan(g, 3e5, n - 2, stats=True)
    S = snps.astype(np.float64)
    den = np.einsum("ij,ij->i", S @ A, S)
    dot = S @ w
    assert rel(out["den"], den) < 1e-7
    assert np.max(np.abs(out["dot"] - dot) / (np.abs(S) @ np.abs(w))) < 1e-13
    assert np.array_equal(out["sum"], S.sum(1))
    rss = 3e5 - dot * dot / den
    F = (3e5 / rss - 1) * (n - 2)
    assert rel(out["ps"], orc.f_sf(np.maximum(F, 0), 1, n - 2)) < 1e-6
    g.close()

"""Round-4 surface closures, written once and run against the numpy stand-in context on the CPU (tests/test_host_logic.py)
and against the device (tests/test_gpu_round4.py): every check is against numbers the REFERENCE produced
(tests/golden/extras3_n150.npz, extras_n150.npz; generator: tests/golden/make_golden.py:run_extras3)."""
import numpy as np


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


def fast_f_test_with_betas(ctx, ex3, tol=1e-6):
    """LinearModel.fast_f_test(with_betas=True) (linear_models.py:196-257): the residual regressed on [X, s] per SNP; the
    last SNP is monomorphic (rank-deficient design: rss = h0_rss, betas = h0_betas, :236-239)."""
    from mixmogam_amd import linear_models as lm
    n = int(ex3["n"])
    sub = np.vstack([ex3["snps"][:79], np.ones((1, n), dtype=np.int8)])
    lin = lm.LinearModel(list(ex3["y"]), ctx=ctx)
    lin.add_factor(ex3["cofs"][0])
    r = lin.fast_f_test(sub, with_betas=True, Z=np.eye(n))          # Z: accepted and ignored, as in the reference
    # SNP 79 is collinear with the intercept.  The reference's lstsq on the singular design [1, cof, 1] does not report the
    # rank deficiency in either mode (the fixture holds betas of 1e14 and an rss BELOW h0_rss for it): its numbers there are
    # LAPACK rounding, not a result, so the comparison covers the 79 proper SNPs and SNP 79 is checked against the rule the
    # reference states (:236-239: no predictability -> the null model's numbers stay)
    for k in ("ps", "f_stats", "rss", "var_perc"):
        assert rel(r[k][:79], ex3["dbl_lmwb_" + k][:79]) < tol, k
    assert rel(r["h0_rss"], ex3["dbl_lmwb_h0_rss"]) < 1e-10 and rel(r["h0_betas"], ex3["dbl_lmwb_h0_betas"]) < 1e-9
    got, want = np.asarray(r["betas"][:79]), ex3["dbl_lmwb_betas"][:79]
    assert got.shape == want.shape == (79, 3)
    assert np.max(np.abs(got - want)) < tol * max(1.0, np.max(np.abs(want)))
    assert abs(ex3["dbl_lmwb_betas"][79]).max() > 1e6                  # what the reference makes of the singular design
    assert r["rss"][79] == r["h0_rss"][0] and r["ps"][79] == 1.0 and list(r["betas"][79]) == list(r["h0_betas"])
    plain = lin.fast_f_test(sub)
    assert rel(plain["ps"], r["ps"]) < 1e-12 and "betas" not in plain


def transformed_snps_with_betas(ctx, ex3, tol_t=1e-7, tol=1e-6):
    """_emmax_f_test_(with_betas=True, return_transformed_snps=True): under with_betas the reference's M is H' itself
    (:1305-1306), so t_snps are the unprojected rotated SNPs; betas per SNP from lstsq([h0_X, H s], r) (:1323-1326)."""
    from mixmogam_amd import linear_models as lm
    lmm = lm.LinearMixedModel(list(ex3["y"]), ctx=ctx)
    lmm.add_random_effect(ex3["ibs_scaled"])
    lmm.add_factor(ex3["cofs"][0])
    r = lmm._emmax_f_test_(list(ex3["snps"][:64]), ex3["dbl_twb_H"], return_transformed_snps=True, with_betas=True, emma_num=0)
    ref = ex3["dbl_twb_snps"]
    assert np.max(np.abs(np.asarray(r["t_snps"]) - ref)) < tol_t * np.max(np.abs(ref))
    assert rel(r["ps"], ex3["dbl_twb_ps"]) < tol
    want = ex3["dbl_twb_betas"]
    assert np.max(np.abs(np.asarray(r["betas"]) - want)) < tol * max(1.0, np.max(np.abs(want)))


def emmax_multi_four_cofactors(ctx, ex3, tol=1e-6):
    """emmax_multi with q = 5 fixed-effect columns (round 3 stopped at 4) against the reference's loop of emmax() runs."""
    from mixmogam_amd import linear_models as lm
    r = lm.emmax_multi(ex3["snps"], ex3["ys"], ex3["ibs_scaled"], cofactors=[list(c) for c in ex3["cofs"]], ctx=ctx)
    assert r["ps"].shape == ex3["dbl_mc4_ps"].shape
    assert rel(r["ps"], ex3["dbl_mc4_ps"]) < tol
    assert rel(r["delta"], ex3["dbl_mc4_delta"]) < 1e-6


def ml_without_an_eigendecomposition(ctx, ex, tol=1e-6):
    """get_ML (linear_models.py:672-683) on the eigendecomposition-free route: log|K + delta I| and tr (K + delta I)^-1 in
    place of the sums over eigh(K) -- against the reference's own get_ML numbers."""
    from mixmogam_amd import linear_models as lm
    lmm = lm.LinearMixedModel(list(ex["multi_ys"][2]), ctx=ctx)
    lmm.add_random_effect(ex["ibs_scaled"])
    res = lmm.get_estimates_eigen_free(ngrids=100, method='ML')
    res.pop("reml").close()
    for k in ("max_ll", "delta", "ve", "vg", "pseudo_heritability"):
        assert rel(res[k], ex["dbl_ml_" + k]) < tol, (k, res[k], ex["dbl_ml_" + k])


def ibd_kinship_from_normalised_snps(ctx, coll=None):
    """hdf5_data.calculate_ibd_kinship on a tree with pre-normalised float `snps` datasets (/root/reference/hdf5_data.py:37-44;
    one chromosome without one is standardised on the fly, :40-43) against float64 numpy."""
    from mixmogam_amd import hdf5_data, kinship
    rng = np.random.RandomState(8)
    n = 70
    raw = [(rng.random_sample((m, n)) < rng.uniform(0.15, 0.85, size=(m, 1))).astype(np.int8) for m in (130, 90)]
    raw = [r[r.std(1) > 0] for r in raw]
    z = [(r - r.mean(1, keepdims=True)) / r.std(1, keepdims=True) for r in raw]
    tree = {"chr1": {"raw_snps": raw[0], "snps": z[0].astype(np.float32)}, "chr2": {"raw_snps": raw[1]}}
    k, n_snps = hdf5_data.calculate_ibd_kinship(tree, n_indivs=n, chunk_size=50, ctx=ctx, coll=coll)
    zz = np.vstack([z[0].astype(np.float32).astype(np.float64), z[1]])
    want = kinship.scale_k(zz.T @ zz / len(zz))
    assert n_snps == len(zz)
    assert np.max(np.abs(k - want)) < 1e-10 * np.max(np.abs(want))

"""GPU tests of round 5: the permutation tests on the Cholesky square root L^-1 of (K + delta I)^-1 taken from the REML workspace
in HBM (no eigendecomposition) against the oracle fed the SAME matrix; the workspace's L^-1 primitives; the sticky failure
state of the kinship accumulator."""
import numpy as np
import pytest

from conftest import load_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from mixmogam_amd import _lib
    return _lib.get_context()


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


def structured(n, m, seed):
    rng = np.random.RandomState(seed)
    pops = rng.randint(0, 3, size=n)
    freqs = rng.uniform(0.1, 0.9, size=(m, 3))
    snps = (rng.random_sample((m, n)) < freqs[:, pops]).astype(np.int8)
    snps = snps[snps.std(1) > 0]
    y = snps[:5].astype(float).sum(0) + rng.randn(n)
    return snps, y, rng


@pytest.mark.parametrize("n", [150, 300, 1001])
def test_workspace_linv_is_a_square_root_of_the_inverse(ctx, n):
    """mmg_reml_linv_fetch / _apply: H = L^-1 (lower triangular) with H'H = (K + delta I)^-1; H V and H'V as products."""
    from mixmogam_amd import kinship
    snps, y, rng = structured(n, 1500, 11)
    K = kinship.calc_ibs_kinship(snps, ctx=ctx)
    X = np.column_stack([np.ones(n), rng.randn(n)])
    reml = ctx.reml(K, X, y)
    try:
        for delta in (0.37, 4.5):
            H = reml.linv(delta)
            assert np.allclose(H, np.tril(H))
            V = K + delta * np.eye(n)
            assert np.max(np.abs(H.T @ H @ V - np.eye(n))) < 1e-9
            W = rng.randn(n, 3)
            assert rel(reml.linv_apply(delta, W), H @ W) < 1e-10
            assert rel(reml.linv_apply(delta, W[:, 0], trans=True), H.T @ W[:, 0]) < 1e-10
    finally:
        reml.close()


@pytest.mark.parametrize("name", ["struct_n150_s0", "struct_n300_s2"])
def test_permutation_thresholds_on_the_cholesky_root_vs_the_oracle(ctx, name):
    """_emmax_permutations_'s arithmetic (:1125-1175) with H_sqrt_inv := L^-1 from the device (perm_prepare(reml=...) +
    mmg_perm_plan_create_from_reml) against oracle.perm_closed fed the same L^-1: the null fit to 1e-10, max F per permutation to
    1e-7 (the statistic's GEMM runs on four 7-bit digit planes of W' = Ys'H: ~1e-8)."""
    from mixmogam_amd import linear_models as lm
    from oracle import emmax_oracle as orc
    case = load_case(name)
    n = int(case["n"])
    K = case["dbl_ibs_scaled"]
    y = np.asarray(case["y"], dtype=np.float64)
    idx = np.array([np.random.RandomState(100 + p).permutation(n) for p in range(24)])
    lmm = lm.LinearMixedModel(list(y), ctx=ctx)
    lmm.add_random_effect(K)
    est = lmm.get_estimates_eigen_free()
    reml, delta = est["reml"], est["delta"]
    try:
        H = reml.linv(delta)
        lmm_p = lm.LinearMixedModel(list(y), ctx=ctx)
        lmm_p.add_random_effect(K)
        pp = lmm_p.perm_prepare(None, num_perm=len(idx), perm_idx=idx, reml=reml, delta=delta)
        plan = reml.perm_plan(delta, pp["Ys"], pp["h0_rss"])
    finally:
        reml.close()
    g = ctx.geno(case["snps"])
    try:
        min_rss = plan.run(g)
    finally:
        plan.close()
        g.close()
    opp = orc.perm_prepare(y, np.ones((n, 1)), H, idx)
    ref = orc.perm_closed(case["snps"], opp)
    assert rel(pp["h0_rss"], opp["h0_rss"]) < 1e-10
    max_f = (pp["h0_rss"] / min_rss - 1.0) * pp["n_p"]
    assert rel(max_f, ref["max_f_stats"]) < 1e-7


def test_public_permutation_test_without_an_eigendecomposition(ctx):
    """emmax_perm_test (:1819-1841) with no H_sqrt_inv handed in: REML eigendecomposition-free, H = L^-1, C H formed on the device
    (flags bit 1) -- against oracle.perm_public fed the same L^-1; and MMG_PERM_H=eigen keeps the eigendecomposition's matrix."""
    import os
    from mixmogam_amd import linear_models as lm
    from oracle import emmax_oracle as orc
    case = load_case("struct_n300_s3")
    n = int(case["n"])
    K = case["dbl_ibs_scaled"]
    y = np.asarray(case["y"], dtype=np.float64)
    snps = case["snps"][:400]
    idx = np.array([np.random.RandomState(7 + p).permutation(n) for p in range(20)])
    res = lm.emmax_perm_test(snps, list(y), K, num_perm=len(idx), perm_idx=idx, ctx=ctx)
    lmm = lm.LinearMixedModel(list(y), ctx=ctx)
    lmm.add_random_effect(K)
    est = lmm.get_estimates_eigen_free()
    try:
        H = est["reml"].linv(est["delta"])
    finally:
        est["reml"].close()
    want = orc.perm_public(snps, y, np.ones((n, 1)), H, idx, reference_indexing=False)
    assert rel(res["max_f_stats"], want["max_f_stats"]) < 1e-7
    assert "threshold_05" in res
    os.environ["MMG_PERM_H"] = "eigen"
    try:
        lit = lm.emmax_perm_test(snps, list(y), K, num_perm=len(idx), perm_idx=idx, ctx=ctx)
    finally:
        del os.environ["MMG_PERM_H"]
    eo = orc.get_estimates(y, np.ones((n, 1)), orc.scale_k(K))
    want_lit = orc.perm_public(snps, y, np.ones((n, 1)), eo["H_sqrt_inv"], idx, reference_indexing=False)
    # (the eigenvectors' signs are LAPACK's on either side: the statistics agree when they happen to, the distribution always)
    assert np.all(np.isfinite(lit["max_f_stats"])) and lit["max_f_stats"].shape == want_lit["max_f_stats"].shape


def test_run_emmax_perm_takes_the_cholesky_root_and_matches_its_oracle(ctx):
    """hdf5_data.run_emmax_perm at N > EIGEN_FREE_MIN_N: eigendecomposition-free route (timings['route']), thresholds equal the
    oracle's on the same L^-1 and do not depend on the chunking (after-scan form on H'Q = L^-T Q)."""
    from mixmogam_amd import hdf5_data, kinship, linear_models as lm
    from oracle import emmax_oracle as orc
    n = 301
    snps, y, rng = structured(n, 900, 5)
    half = len(snps) // 2
    tree = {"c1": {"raw_snps": snps[:half], "freqs": snps[:half].mean(1), "positions": np.arange(half)},
            "c2": {"raw_snps": snps[half:], "freqs": snps[half:].mean(1), "positions": np.arange(len(snps) - half)}}
    idx = np.array([rng.permutation(n) for _ in range(16)])
    tm = {}
    a = hdf5_data.run_emmax_perm(tree, None, min_maf=None, chunk_size=97, num_perm=len(idx), perm_idx=idx, ctx=ctx, phenotypes=y)
    b = hdf5_data.run_emmax(tree, y, min_maf=None, chunk_size=10 ** 6, num_perm=len(idx), perm_idx=idx, ctx=ctx, timings=tm)
    assert "eigendecomposition-free" in tm["route"]
    assert rel(a["perm_max_f_stats"], b["perm_max_f_stats"]) < 1e-7
    # the oracle on the same square root: kinship as the driver computes it, delta as the driver found it
    Kg = np.asarray(b["kinship"])
    delta = 1.0 / b["pseudo_heritability"] - 1.0
    H = np.linalg.inv(np.linalg.cholesky(kinship.scale_k(Kg) + delta * np.eye(n)))
    pp = orc.perm_prepare(y, np.ones((n, 1)), H, idx)
    want = orc.perm_closed(snps[:half], pp)                      # the test runs on every chromosome but the last (:294-311)
    assert rel(b["perm_max_f_stats"], want["max_f_stats"]) < 1e-6


def test_rejected_grm_call_leaves_the_accumulator_as_it_was(ctx):
    """A call that is rejected before it adds (monomorphic SNP) leaves the accumulator as it was; the sticky failure state itself
    (a launch error mid-call) cannot be provoked from outside -- its plumbing is covered by the error text of every entry point."""
    from mixmogam_amd import _lib
    rng = np.random.RandomState(2)
    n = 100
    good = (rng.random_sample((300, n)) < 0.4).astype(np.int8)
    good = good[good.std(1) > 0]
    bad = good.copy()
    bad[7] = 1
    acc = ctx.kinship_accumulator(n)
    g1, g2 = ctx.geno(good), ctx.geno(bad)
    try:
        acc.add_grm(g1)
        with pytest.raises(_lib.MixmogamHipError, match="monomorphic"):
            acc.add_grm(g2)
        acc.add_grm(g1)                                            # still usable: nothing of the rejected call was added
        K, cnt = acc.fetch()
        assert cnt == 2 * len(good)
        z = (good - good.mean(1, keepdims=True)) / good.std(1, keepdims=True)
        assert np.max(np.abs(K - 2 * z.T @ z)) / np.max(np.abs(K)) < 1e-8
    finally:
        acc.close(); g1.close(); g2.close()


def test_ibs_kinship_kept_in_hbm_feeds_emmax_without_a_host_visit(ctx):
    """kinship.calc_ibs_kinship(keep_device=True) (mmg_kin_acc_set_ibs): the same matrix as the host-returning call (bit for bit
    above 2048 individuals, where both are formed on the device; to 1e-13 below), and emmax() on it equals emmax() on the array."""
    from mixmogam_amd import kinship, linear_models as lm
    for n, m in ((300, 1200), (2100, 900)):
        snps, y, _rng = structured(n, m, 21)
        Kh = kinship.calc_ibs_kinship(snps, ctx=ctx)
        Kd = kinship.calc_ibs_kinship(snps, ctx=ctx, keep_device=True)
        try:
            got = np.asarray(Kd)
            assert got.shape == (n, n)
            assert np.max(np.abs(got - Kh)) <= (0.0 if n > 2048 else 1e-13)
            a = lm.emmax(snps[:500], list(y), Kh, ctx=ctx)
            b = lm.emmax(snps[:500], list(y), Kd, ctx=ctx)
            assert rel(b["ps"], a["ps"]) < 1e-9 and abs(b["pseudo_heritability"] - a["pseudo_heritability"]) < 1e-10
        finally:
            Kd.close()


def test_get_emma_reml_estimates_is_lazy_about_the_matrices(ctx):
    """get_emma_reml_estimates above EIGEN_FREE_MIN_N: scalars, beta and the transformed X / Y without an eigendecomposition
    (golden values of the reference); H_sqrt_inv (the Cholesky root: H'H = (K + delta I)^-1) and eig_L only when asked for."""
    from mixmogam_amd import linear_models as lm
    case = load_case("struct_n300_s2")
    res = lm.get_emma_reml_estimates(list(case["y"]), case["dbl_ibs_scaled"], cofactors=case["cof"], ctx=ctx)
    assert isinstance(res, lm._LazyEstimates) and not dict.__contains__(res, "eig_L") and not dict.__contains__(res, "H_sqrt_inv")
    for k in ("max_ll", "delta", "ve", "vg", "pseudo_heritability"):
        assert rel(res[k], case["dbl_reml_" + k]) < 1e-7, k
    assert rel(res["beta"], case["dbl_reml_beta"]) < 1e-6
    H = res["H_sqrt_inv"]
    assert np.allclose(H, np.tril(H))
    probe = np.random.RandomState(99).randn(len(case["y"]), 3)
    assert rel(H.T @ (H @ probe), case["dbl_HtH_probe"]) < 1e-6
    assert rel(H @ res["lmm"].X, res["X_t"]) < 1e-9 and rel(H @ res["lmm"].Y.reshape(-1), res["Y_t"].reshape(-1)) < 1e-9
    assert np.max(np.abs(res["eig_L"]["values"] - case["dbl_eig_L_values"])) < 1e-9
    res.close()

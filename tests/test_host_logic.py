"""-m "not gpu": the HOST side of the product (mixmogam_amd.linear_models / kinship / hdf5_data) driven
through a numpy stand-in for the device context (tests/fake_ctx.py), against the reference's golden
vectors.  Covers what runs on the host in production: model building, REML grid + secant search, the
closed-form scan preparation, with_betas algebra, MLMM stepping and the chunked drivers."""
import numpy as np
import pytest

from conftest import load_case
from fake_ctx import FakeContext
from mixmogam_amd import hdf5_data, kinship
from mixmogam_amd import linear_models as lm


def rel(a, b):
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


@pytest.fixture
def ctx():
    return FakeContext()


def test_emmax_and_reml_host_logic_vs_golden(case, ctx):
    res = lm.emmax(list(case["snps"]), list(case["y"]), case["dbl_ibs_scaled"], cofactors=case["cof"], ctx=ctx)
    assert rel(res["ps"], case["dbl_emmax_ps"]) < 1e-7
    assert rel(res["h0_rss"], case["dbl_emmax_h0_rss"]) < 1e-9
    for k in ("pseudo_heritability", "ve", "vg", "max_ll"):
        assert rel(res[k], case["dbl_emmax_" + k]) < 1e-8, k
    reml = lm.get_emma_reml_estimates(list(case["y"]), case["dbl_ibs_scaled"], cofactors=case["cof"], ctx=ctx)
    for k in ("max_ll", "delta", "ve", "vg", "pseudo_heritability"):
        assert rel(reml[k], case["dbl_reml_" + k]) < 1e-8, k
    assert np.max(np.abs(reml["eig_L"]["values"] - case["dbl_eig_L_values"])) < 1e-10


def test_with_betas_and_kinship_host_logic(case, ctx):
    lmm = lm.LinearMixedModel(list(case["y"]), ctx=ctx)
    lmm.add_random_effect(case["dbl_ibs_scaled"])
    if case["cof"] is not None:
        for c in case["cof"]:
            lmm.add_factor(c)
    wb = lmm.emmax_f_test(case["snps"][:200], with_betas=True, emma_num=0)
    assert np.max(np.abs(np.asarray(wb["betas"]) - case["dbl_wb_betas"])) < 1e-7 * max(1, np.abs(case["dbl_wb_betas"]).max())
    assert rel(kinship.calc_ibs_kinship(list(case["snps"]), ctx=ctx), case["dbl_ibs_scaled"]) < 1e-13
    assert np.max(np.abs(kinship.calc_ibd_kinship(case["snps"], ctx=ctx) - case["dbl_ibd_scaled"])) < 1e-11


def test_add_factor_rejects_dependent_cofactor(ctx):
    case = load_case("struct_n150_s0")
    lmm = lm.LinearMixedModel(list(case["y"]), ctx=ctx)
    with pytest.warns(UserWarning):
        assert lmm.add_factor(np.full(150, 3.0)) is False          # collinear with the intercept (:105-108)
    assert lmm.add_factor(case["snps"][0]) is True and lmm.X.shape == (150, 2)


def test_mlmm_host_logic_vs_golden(ctx):
    case = load_case("struct_n150_s0")
    m = len(case["snps"])
    res = lm.mlmm(list(case["y"]), case["dbl_ibs_scaled"], num_steps=3, forward_backwards=True,
                  snps=case["snps"], positions=list(range(m)), chromosomes=[1] * m, ctx=ctx)
    want = case["dbl_mlmm_steps"]
    got = np.array([[np.nan if si[k] is None else float(np.asarray(si[k]).reshape(-1)[0])
                     for k in ("pseudo_heritability", "ll", "bic", "e_bic", "m_bic")] for si in res["step_info_list"]])
    assert np.allclose(got, want[:, :5], rtol=1e-6)
    assert [[c[1] for c in si["cofactors"]] for si in res["step_info_list"]] == \
        [[p for p in row if p >= 0] for row in case["dbl_mlmm_cof_pos"]]
    for c in ("ebics", "mbics", "bonf", "mbonf", "min_cof_ppa"):
        assert res["opt_dict"][c] == int(case["dbl_mlmm_opt_" + c])


def test_chunked_driver_host_logic(ctx):
    rng = np.random.RandomState(3)
    n = 120
    snps = (rng.random_sample((700, n)) < rng.uniform(0.05, 0.95, size=(700, 1))).astype(np.int8)
    snps = snps[(snps.sum(1) > 0) & (snps.sum(1) < n)]
    src = {"c1": {"raw_snps": snps[:400], "freqs": snps[:400].mean(1), "positions": np.arange(400)},
           "c2": {"raw_snps": snps[400:], "freqs": snps[400:].mean(1), "positions": np.arange(len(snps) - 400)}}
    y = rng.randn(n) + snps[3]
    a = hdf5_data.run_emmax(src, y, min_maf=0.1, chunk_size=64, ctx=ctx)
    b = hdf5_data.run_emmax(src, y, min_maf=0.1, chunk_size=10 ** 6, ctx=ctx)
    for c in ("c1", "c2"):
        assert rel(a["chrom_results"][c]["ps"], b["chrom_results"][c]["ps"]) < 1e-9
    keep = np.minimum(snps.mean(1), 1 - snps.mean(1)) > 0.1
    assert a["num_snps"] == int(keep.sum())
    from oracle import emmax_oracle as orc
    ref = orc.emmax(snps[keep], y, a["kinship"])
    got = np.concatenate([a["chrom_results"][c]["ps"] for c in ("c1", "c2")])
    assert rel(got, ref["ps"]) < 1e-7


@pytest.mark.parametrize("name", ["struct_n150_s0", "struct_n300_s3", "bern_n200_s4"])
def test_reml_sums_from_eig_L_equal_the_eig_R_route(name):
    """get_estimates takes the four likelihood sums from eig_L alone (no second N^3 eigh); the reference's
    own route through eig_R (kept as _get_estimates_with) gives the same delta, likelihood and variance
    components to rounding, for REML and ML, with and without cofactors, also with an extra SNP column (xs)."""
    from mixmogam_amd import linear_models as lm
    case = load_case(name)
    m = lm.LinearMixedModel(list(case["y"]), ctx=FakeContext())
    m.add_random_effect(case["dbl_ibs_scaled"])
    if case["cof"] is not None:
        for c in case["cof"]:
            m.add_factor(c)
    eig_L = m._get_eigen_L_()
    for method in ("REML", "ML"):
        fast = m.get_estimates(eig_L, method=method)
        ref = m._get_estimates_with(eig_L, m._get_eigen_R_(X=m.X), method)
        for k in ("delta", "max_ll", "vg", "ve", "pseudo_heritability"):
            assert abs(fast[k] / ref[k] - 1) < 1e-10, (method, k)
        assert np.allclose(fast["H_sqrt_inv"], ref["H_sqrt_inv"], rtol=1e-10, atol=1e-12)
    xs = case["snps"][7].astype(np.float64).reshape(-1, 1)
    fast = m.get_estimates(eig_L, xs=xs, return_f_stat=True, return_pvalue=True)
    X = np.hstack([m.X, xs])
    ref = m.get_estimates(eig_L, xs=xs, return_f_stat=True, return_pvalue=True, eig_R=m._get_eigen_R_(X=X))
    for k in ("delta", "max_ll", "f_stat", "p_val"):
        assert abs(fast[k] / ref[k] - 1) < 1e-9, k


def test_emmax_multi_host_logic_vs_reference_loop(ctx):
    """emmax_multi (one eigh, rotated SNPs, per-phenotype REML in the eigenbasis) == the reference's loop of emmax()
    runs over six phenotypes with six different variance ratios, without and with a shared cofactor."""
    from conftest import load_extras
    ex = load_extras()
    for tag, cof in (("multi", None), ("multic", [ex["multi_cof"]])):
        res = lm.emmax_multi(list(ex["snps"]), ex["multi_ys"], ex["ibs_scaled"], cofactors=cof, ctx=ctx)
        assert res["ps"].shape == ex["dbl_%s_ps" % tag].shape
        assert rel(res["ps"], ex["dbl_%s_ps" % tag]) < 1e-7
        assert rel(res["rss"], ex["dbl_%s_rss" % tag]) < 1e-8
        assert np.max(np.abs(res["var_perc"] - ex["dbl_%s_var_perc" % tag])) < 1e-9
        for k in ("h0_rss", "pseudo_heritability", "max_ll"):
            assert rel(res[k], ex["dbl_%s_%s" % (tag, k)]) < 1e-8, k
        # a small store budget forces several rotate + scan rounds over SNP chunks: same numbers
        res2 = lm.emmax_multi(list(ex["snps"]), ex["multi_ys"], ex["ibs_scaled"], cofactors=cof, ctx=ctx,
                              max_store_bytes=256 * 8 * 192)
        assert np.array_equal(res2["ps"], res["ps"])

"""GPU tests of round 6: the HDF5 DRIVERS (hdf5_data.calculate_ibd_kinship / run_emmax / run_emmax_perm) against what the
reference's own hdf5_data.py wrote to its files (tests/golden/hdf5_n200.npz, produced by make_golden.run_hdf5 under
python3.9 + h5py); the N = 1000 and config-1-shaped reference runs; small-N routes at sizes that are not multiples of 64."""
import os

import numpy as np
import pytest

from conftest import load_hdf5_golden, reference_row_signs

pytestmark = pytest.mark.gpu

orc = pytest.importorskip("oracle.emmax_oracle")


@pytest.fixture(scope="module")
def ctx():
    from mixmogam_amd import _lib
    return _lib.get_context()


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64).reshape(-1), np.asarray(b, dtype=np.float64).reshape(-1)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))) if len(a) else 0.0


@pytest.fixture(scope="module")
def h5gold():
    return load_hdf5_golden()


def _container(tmp_path, d, variant, name="geno.mmg", packed_bits=0):
    """The reference's genotype file rebuilt from the fixture's arrays (plink2hdf5.py:27-28,111-118,226 layout)."""
    from mixmogam_amd import chunkstore
    chroms = d[variant + "_chroms"]
    return chunkstore.write_genotype_container(
        str(tmp_path / name), {c: s for c, s, _f, _p in chroms}, d[variant + "_indiv_ids"],
        phenotypes=d[variant + "_phenotypes"], positions={c: p for c, _s, _f, p in chroms},
        freqs={c: f for c, _s, f, _p in chroms}, packed_bits=packed_bits)


def _read(path):
    from mixmogam_amd import chunkstore
    return chunkstore.open_container(path, "r")


@pytest.mark.parametrize("variant", ["bin", "dip"])
def test_calculate_ibd_kinship_driver_vs_the_reference_run(ctx, tmp_path, h5gold, variant):
    """hdf5_data.py:17-62: every SNP of every chromosome (no MAF filter), K stored in the genotype file as 'kinship';
    a second call leaves it; overwrite=True recomputes.  <= 1e-9 of the double-promoted run (products are exact here:
    4-plane int8 GRM), and within the literal run's own fp32 accumulator distance."""
    from mixmogam_amd import hdf5_data
    d = h5gold
    path = _container(tmp_path, d, variant)
    chunk = int(d["chunk_size"])
    k, n_snps = hdf5_data.calculate_ibd_kinship(path, chunk_size=chunk, ctx=ctx)
    assert n_snps == sum(len(s) for _c, s, _f, _p in d[variant + "_chroms"])
    stored = np.asarray(_read(path)["kinship"][...])
    assert np.array_equal(stored, k)
    assert np.abs(k - d[variant + "_dbl_calc_kinship"]).max() < 1e-9
    assert np.abs(k - d[variant + "_lit_calc_kinship"]).max() < 2e-5
    again, none = hdf5_data.calculate_ibd_kinship(path, chunk_size=chunk, ctx=ctx)      # 'kinship already there.' (:61-62)
    assert none is None and np.array_equal(again, k)
    k2, _ = hdf5_data.calculate_ibd_kinship(path, chunk_size=400, overwrite=True, ctx=ctx)
    assert np.abs(k2 - k).max() < 1e-12                                                 # chunking is invisible
    if variant == "bin":
        # a file that carries pre-normalised `snps` datasets: taken as they are (:36-38), against the reference's run on such a file
        from mixmogam_amd import chunkstore
        path_n = _container(tmp_path, d, variant, name="geno_norm.mmg")
        st = chunkstore.open_container(path_n, "a")
        for c, s, _f, _p in d[variant + "_chroms"]:
            x = s.astype(np.float64)
            st["genot_data"][c].create_dataset("snps", data=(x - x.mean(1, keepdims=True)) / x.std(1, keepdims=True))
        st.flush()
        st.close()
        kn, n_n = hdf5_data.calculate_ibd_kinship(path_n, chunk_size=chunk, ctx=ctx)
        assert n_n == n_snps
        assert np.abs(kn - d["bin_dbl_calcnorm_kinship"]).max() < 1e-9
        assert np.abs(kn - d["bin_lit_calcnorm_kinship"]).max() < 2e-5


@pytest.mark.parametrize("variant,packed_bits", [("bin", 0), ("dip", 0), ("bin", 1), ("dip", 2)])
def test_run_emmax_driver_vs_the_reference_run(ctx, tmp_path, h5gold, variant, packed_bits):
    """hdf5_data.py:70-187 through the files, as the reference is called: MAF filter `mafs > min_maf` (:91-93, two rows
    sit ON the threshold), kinship from the kept rows, REML once, scan per chromosome; result file datasets (:142-184)
    equal to the reference's own -- kept positions identical, p-values <= 1e-6 of the double-promoted run."""
    from mixmogam_amd import hdf5_data
    d = h5gold
    ref = lambda k: d["%s_dbl_%s" % (variant, k)]
    path = _container(tmp_path, d, variant, packed_bits=packed_bits)
    out_file = str(tmp_path / "res.mmg")
    res = hdf5_data.run_emmax(path, out_file, min_maf=float(d["min_maf"]), recalculate_kinship=True,
                              chunk_size=int(d["chunk_size"]), ctx=ctx)
    o = _read(out_file)
    assert list(o["chrom_results"].keys()) == list(ref("emmax_chroms"))
    assert int(np.asarray(o["num_snps"][...])) == int(ref("emmax_num_snps"))              # :150: the input file's total
    for key in ("pseudo_heritability", "ve", "vg", "max_ll"):
        assert rel(np.asarray(o[key][...]), ref("emmax_" + key)) < 1e-6, key
    for c in ref("emmax_chroms"):
        assert np.array_equal(np.asarray(o["chrom_results"][c]["positions"][...]), ref("emmax_%s_positions" % c))
        assert rel(np.asarray(o["chrom_results"][c]["ps"][...]), ref("emmax_%s_ps" % c)) < 1e-6
        # the literal (fp32) reference run is as far from this as from its own double-promoted self
        lit = d["%s_lit_emmax_%s_ps" % (variant, c)]
        assert rel(np.asarray(o["chrom_results"][c]["ps"][...]), lit) < 1.5 * max(rel(lit, ref("emmax_%s_ps" % c)), 1e-4)
    assert np.abs(res["kinship"] - d[variant + "_dbl_perm_kinship"]).max() < 1e-9        # the kept rows' kinship (:82-111)
    assert np.abs(res["kinship"] - d[variant + "_lit_perm_kinship"]).max() < 2e-5


@pytest.mark.parametrize("variant", ["bin", "dip"])
def test_run_emmax_with_the_stored_kinship_vs_the_reference_run(ctx, tmp_path, h5gold, variant):
    """recalculate_kinship=False (:113-115): the kinship calculate_ibd_kinship left in the genotype file (all SNPs)."""
    from mixmogam_amd import hdf5_data
    d = h5gold
    path = _container(tmp_path, d, variant)
    with pytest.raises(AssertionError):                                                  # 'Kinship is missing.' (:114)
        hdf5_data.run_emmax(path, str(tmp_path / "x.mmg"), recalculate_kinship=False, ctx=ctx)
    hdf5_data.calculate_ibd_kinship(path, chunk_size=int(d["chunk_size"]), ctx=ctx)
    out_file = str(tmp_path / "res_k.mmg")
    hdf5_data.run_emmax(path, out_file, min_maf=float(d["min_maf"]), recalculate_kinship=False, ctx=ctx)
    o = _read(out_file)
    assert rel(np.asarray(o["pseudo_heritability"][...]), d["%s_dbl_storedk_pseudo_heritability" % variant]) < 1e-6
    for c in d["%s_dbl_emmax_chroms" % variant]:
        assert rel(np.asarray(o["chrom_results"][c]["ps"][...]), d["%s_dbl_storedk_%s_ps" % (variant, c)]) < 1e-6


@pytest.mark.parametrize("variant", ["bin", "dip"])
def test_run_emmax_perm_driver_vs_the_reference_run(ctx, tmp_path, h5gold, variant):
    """hdf5_data.py:191-351 replayed: the recorded shuffles in the reference's own H_sqrt_inv (row signs are LAPACK's
    choice and decide each permutation's outcome -- reference_row_signs rebuilds its matrix from this K and delta and
    verifies it against recorded products).  The result FILE equals the reference's: kinship, scan p-values, sorted
    perm_min_ps / perm_max_f_stats (all chromosomes but the last, :294-311), the 5 % entries at index num_perm // 20 of each
    (:342-347), and num_snps = the last chromosome's kept count (:253,289)."""
    from mixmogam_amd import hdf5_data, kinship
    d = h5gold
    tag = variant + "_dbl"
    ref = lambda k: d["%s_%s" % (tag, k)]
    path = _container(tmp_path, d, variant)
    nperm, min_maf, chunk = int(d["num_perm"]), float(d["min_maf"]), int(d["chunk_size"])
    # this model's H_sqrt_inv (any signs) -> the reference's signs
    y, n = d[variant + "_phenotypes"], int(d["n"])
    est = orc.get_estimates(y, np.ones((n, 1)), kinship.scale_k(d[tag + "_perm_kinship"]))
    H_ref = reference_row_signs(d, tag, est["H_sqrt_inv"])
    out_file = str(tmp_path / "res_perm.mmg")
    res = hdf5_data.run_emmax_perm(path, out_file, min_maf=min_maf, chunk_size=chunk, num_perm=nperm,
                                   perm_idx=ref("perm_idx"), perm_h=H_ref, ctx=ctx)
    o = _read(out_file)
    assert np.abs(np.asarray(o["kinship"][...]) - ref("perm_kinship")).max() < 1e-9
    assert int(np.asarray(o["num_snps"][...])) == int(ref("perm_num_snps"))
    for key in ("pseudo_heritability", "ve", "vg", "max_ll"):
        assert rel(np.asarray(o[key][...]), ref("perm_" + key)) < 1e-6, key
    for c in ref("emmax_chroms"):
        assert rel(np.asarray(o["chrom_results"][c]["ps"][...]), ref("perm_%s_ps" % c)) < 1e-6
    # max F per permutation: the statistic's GEMM runs on 7-bit digit planes (~1e-8); p = f.sf(F) amplifies by ~ln(1/p)
    assert rel(np.asarray(o["perm_max_f_stats"][...]), ref("perm_perm_max_f_stats")) < 1e-6
    assert rel(np.asarray(o["perm_min_ps"][...]), ref("perm_perm_min_ps")) < 1e-5
    assert rel(np.asarray(o["five_perc_perm_min_ps"][...]), ref("perm_five_perc_perm_min_ps")) < 1e-5
    assert rel(np.asarray(o["five_perc_perm_max_f_stats"][...]), ref("perm_five_perc_perm_max_f_stats")) < 1e-6
    assert rel(res["threshold_05"][0], ref("perm_five_perc_perm_min_ps")) < 1e-5
    # without the replay matrix the test runs in this model's own square root: another draw from the same null, so the
    # sorted minima differ by Monte-Carlo noise only -- the 5 % threshold of 40 permutations within a factor of 30
    own = hdf5_data.run_emmax_perm(path, None, min_maf=min_maf, chunk_size=chunk, num_perm=nperm,
                                   perm_idx=ref("perm_idx"), ctx=ctx)
    assert 1 / 30.0 < own["threshold_05"][0] / float(ref("perm_five_perc_perm_min_ps")) < 30.0
    assert rel(own["chrom_results"]["chrom_1"]["ps"], ref("perm_chrom_1_ps")) < 1e-6


# ------------------------------------------------------------------ N = 1000 and config-1 reference runs
def test_exact_emma_refinement_vs_the_n1000_reference_run(ctx):
    """emma_num=10 at N = 1000 with two cofactors (linear_models.py:1365-1377): the ten top hits re-estimated exactly, the
    rest untouched -- against the reference's own run."""
    from conftest import load_case
    from mixmogam_amd import linear_models as lm
    case = load_case("struct_n1000_s7")
    res = lm.emmax(case["snps"], list(case["y"]), case["dbl_ibs_scaled"], cofactors=case["cof"], emma_num=10)
    assert rel(res["ps"], case["dbl_emma10_ps"]) < 1e-5
    top = np.argsort(case["dbl_emmax_ps"], kind="stable")[:10]
    assert rel(res["rss"][top], case["dbl_emma10_rss"][top]) < 1e-6
    assert rel(res["f_stats"][top], case["dbl_emma10_f_stats"][top]) < 1e-5


def test_config1_call_sequence_vs_the_reference_run(ctx):
    """examples.py:72-96 verbatim against the build's modules on config 1's shape -- parse the phenotype file, construct the
    SNP data set, coordinate, calc_ibs_kinship(sd.get_snps()), emmax(sd.get_snps(), phend.get_values(5), K) -- and every
    number against the reference's own run of the same sequence (tests/golden/ft10_config1.npz)."""
    from conftest import GOLDEN, load_case
    from mixmogam_amd import kinship, linear_models as lm, phenotypeData as pd, snpsdata
    c = load_case("ft10_config1")
    phend = pd.parse_phenotype_file(os.path.join(GOLDEN, "at_phenotypes_ft10_ft16.csv"))
    sd = snpsdata.construct_snps_data_set(c["raw_snps"], list(c["positions"]), list(c["chromosomes"]), list(c["accessions"]))
    sd.coordinate_w_phenotype_data(phend, 5)
    K = kinship.calc_ibs_kinship(sd.get_snps())
    assert np.abs(np.asarray(K) - c["dbl_ibs_scaled"]).max() < 1e-13
    res = lm.emmax(sd.get_snps(), phend.get_values(5), K)
    assert rel(res["ps"], c["dbl_emmax_ps"]) < 1e-6
    assert rel(res["rss"], c["dbl_emmax_rss"]) < 1e-8
    for k in ("pseudo_heritability", "ve", "vg", "max_ll"):
        assert rel(res[k], c["dbl_emmax_" + k]) < 1e-7, k
    assert np.argmin(res["ps"]) == np.argmin(c["dbl_emmax_ps"])
    lit = c["lit_emmax_ps"].astype(np.float64)
    ok = lit > 1e-30
    assert np.median(np.abs(res["ps"][ok] / lit[ok] - 1)) < 1e-3          # the literal fp32 run: its own noise


# ------------------------------------------------------------------ small N, not multiples of 64 (advisor r5)
def _small(n, m, seed):
    rng = np.random.RandomState(seed)
    pops = rng.randint(0, 2, size=n)
    freqs = rng.uniform(0.15, 0.85, size=(m, 2))
    snps = (rng.random_sample((m, n)) < freqs[:, pops]).astype(np.int8)
    snps = snps[snps.std(1) > 0]
    y = snps[:3].astype(float).sum(0) + rng.randn(n)
    return snps, y, rng


@pytest.mark.parametrize("n", [16, 17, 63, 65, 100])
def test_small_n_routes_vs_the_oracle(ctx, n):
    """EIGEN_FREE_MIN_N = 15 sends N = 16..255 down the per-delta Cholesky route: REML scalars and the scan through emmax(), the
    L^-1 primitives, the device scan model and the permutation test on L^-1, at sizes that are not multiples of 64."""
    from mixmogam_amd import kinship, linear_models as lm
    snps, y, rng = _small(n, 400, n)
    K = kinship.calc_ibs_kinship(snps, ctx=ctx)
    assert np.array_equal(np.rint((kinship.calc_ibs_kinship(snps, scaled=False, ctx=ctx) - 0.5) * 2 * len(snps)).astype(np.int64),
                          orc.ibs_counts(snps))
    ref = orc.emmax(snps, y, K)
    res = lm.emmax(snps, y, K, ctx=ctx)
    flat = abs(np.log10(ref["delta"])) > 9.5                       # an optimum on the grid's edge: both sit there
    assert rel(res["ps"], ref["ps"]) < (1e-6 if not flat else 1e-5)
    assert rel(res["pseudo_heritability"], ref["pseudo_heritability"]) < 1e-6 or flat
    X = np.ones((n, 1))
    reml = ctx.reml(kinship.scale_k(K), X, y)
    try:
        delta = float(ref["delta"]) if not flat else 0.7
        H = reml.linv(delta)
        V = kinship.scale_k(K) + delta * np.eye(n)
        assert np.allclose(H, np.tril(H)) and np.max(np.abs(H.T @ H @ V - np.eye(n))) < 1e-9
        W = rng.randn(n, 2)
        assert rel(reml.linv_apply(delta, W), H @ W) < 1e-10
        h0_rss, beta = reml.scan_model(delta)
        prep = orc.scan_prepare(y, X, H)
        assert rel(h0_rss, prep["h0_rss"]) < 1e-9
        got = ctx.scan(ctx.geno(snps), h0_rss, n - 2)
        assert rel(got["ps"], orc.scan_closed(snps, prep)["ps"]) < 1e-6
        idx = np.array([np.random.RandomState(50 + p).permutation(n) for p in range(21)])
        lmm = lm.LinearMixedModel(list(y), ctx=ctx)
        lmm.add_random_effect(K)
        pp = lmm.perm_prepare(None, num_perm=len(idx), perm_idx=idx, reml=reml, delta=delta)
        plan = reml.perm_plan(delta, pp["Ys"], pp["h0_rss"])
        try:
            min_rss = plan.run(ctx.geno(snps))
        finally:
            plan.close()
        want = orc.perm_closed(snps, orc.perm_prepare(y, X, H, idx))
        assert rel(min_rss, want["min_rss"]) < 1e-7
    finally:
        reml.close()


def test_reml_estimates_result_holds_no_device_workspace(ctx):
    """get_emma_reml_estimates on the eigendecomposition-free route (advisor r5): the 100-point grid of get_REML (:653), and a
    result that keeps no HBM -- the matrices the reference also returns are rebuilt from the model when first asked for."""
    from conftest import load_case
    from mixmogam_amd import linear_models as lm
    case = load_case("struct_n300_s2")
    out = [lm.get_emma_reml_estimates(list(case["y"]), case["dbl_ibs_scaled"], ctx=ctx) for _ in range(4)]
    for res in out:
        assert isinstance(res, lm._LazyEstimates) and not hasattr(res, "_reml")
        assert rel(res["delta"], case["dbl_reml_delta"]) < 1e-7
    H = out[0]["H_sqrt_inv"]                                          # built now, from a workspace that is closed again
    n = len(case["y"])
    probe = np.random.RandomState(99).randn(n, 3)
    assert rel(H.T @ (H @ probe), case["dbl_HtH_probe"]) < 1e-6
    assert rel(out[1]["X_t"], H @ np.ones((n, 1))) < 1e-9
    lmm = lm.LinearMixedModel(list(case["y"]), ctx=ctx)
    lmm.add_random_effect(case["dbl_ibs_scaled"])
    assert rel(lmm.get_REML()["delta"], case["dbl_reml_delta"]) < 1e-7     # the eigen route's 100-point search agrees


# ------------------------------------------------------------------ 0/1/2 stores: the stacked single-GEMM IBS route
@pytest.mark.parametrize("n,m", [(2100, 3000), (2305, 1111)])
def test_diploid_ibs_device_route_is_one_stacked_gemm_and_bit_exact(ctx, n, m):
    """calc_ibs_kinship('diploid_int') above 2048 individuals runs in HBM (mmg_kinship_ibs_diploid_f64); round 6: both indicator
    images [s >= 1], [s >= 2] from one read of the store, stacked into one 2 M-row FP4 image, ONE GEMM.  Exact counts, so the
    unscaled kinship equals kinship.py:33-41's expression bit for bit (integers / M); the scaled one follows scale_k's rule."""
    from mixmogam_amd import kinship
    rng = np.random.RandomState(n + m)
    f = rng.uniform(0.05, 0.95, size=(m, 1))
    dip = ((rng.random_sample((m, n)) < f).astype(np.int8) + (rng.random_sample((m, n)) < f).astype(np.int8))
    dip[7] = 2
    dip[8] = 0
    u1, u2 = (dip >= 1).astype(np.float64), (dip >= 2).astype(np.float64)
    c12 = u1.T @ u1 + u2.T @ u2                                       # exact in float64 (counts <= 2 M)
    r = np.diag(c12)
    want = (m - 0.5 * (r[:, None] + r[None, :] - 2.0 * c12)) / m
    np.fill_diagonal(want, 1.0)
    g = ctx.geno(dip)
    got = kinship.calc_ibs_kinship(None, snps_data_format='diploid_int', scaled=False, ctx=ctx, geno=g)
    assert np.array_equal(np.asarray(got), want)
    assert ctx.kernel_ms("kinship") > 0 and ctx.kernel_ms("pack") > 0   # one GEMM launch, one image pass were timed
    got_s = kinship.calc_ibs_kinship(None, snps_data_format='diploid_int', scaled=True, ctx=ctx, geno=g)
    assert np.abs(np.asarray(got_s) - orc.scale_k(want)).max() < 1e-12
    g.close()


def test_streamed_two_bit_container_of_diploid_codes(ctx, tmp_path):
    """The coding plink2hdf5.py writes (0/1/2, `freqs` = mean / 2) streamed from a 2-bit packed container at a size where every
    fused path is the large-N one (N = 6000 x M = 120,000, three chromosomes, MAF filter active): exact GRM of the kept rows
    against a float64 host corner, p-values of sampled SNPs against the reference's per-SNP arithmetic in float64
    (linear_models.py:1316-1349) on the product's own kinship."""
    from mixmogam_amd import chunkstore, hdf5_data, linear_models as lm
    rng = np.random.RandomState(66)
    n, sizes = 6000, (50000, 30000, 40000)
    pops = rng.randint(0, 3, size=n)
    chroms, fr = {}, {}
    for ci, m in enumerate(sizes):
        f = rng.uniform(0.03, 0.97, size=(m, 3))[:, pops]
        s = (rng.random_sample((m, n)) < f).astype(np.int8) + (rng.random_sample((m, n)) < f).astype(np.int8)
        chroms["chrom_%d" % (ci + 1)] = s
        fr["chrom_%d" % (ci + 1)] = s.mean(1) / 2.0
    y = chroms["chrom_2"][:5].astype(np.float64).T @ rng.exponential(1.0, 5) + 3.0 * rng.randn(n)
    path = chunkstore.write_genotype_container(str(tmp_path / "dip2.mmg"), chroms, np.arange(n), phenotypes=y, packed_bits=2)
    res = hdf5_data.run_emmax(path, str(tmp_path / "res.mmg"), min_maf=0.1, chunk_size=20000, ctx=ctx)
    keep = {c: np.minimum(fr[c], 1 - fr[c]) > 0.1 for c in chroms}
    kept = np.vstack([chroms[c][keep[c]] for c in chroms])
    assert res["num_snps"] == len(kept) and 0.5 * sum(sizes) < len(kept) < sum(sizes)
    # kinship: a 400 x 400 corner of the GRM of the kept rows in float64, scaled with the product's own factor
    K = np.asarray(res["kinship"])
    z = kept[:, :400].astype(np.float64)
    mu, sd = kept.mean(1, dtype=np.float64), kept.std(1, dtype=np.float64)
    z = (z - mu[:, None]) / sd[:, None]
    corner = z.T @ z / len(kept)
    scale = K[0, 0] / corner[0, 0]
    assert np.abs(K[:400, :400] - scale * corner).max() < 1e-8 * np.abs(K).max()
    assert abs((np.trace(K) - K.sum() / n) - (n - 1)) < 1e-6 * n           # scale_k's normalisation
    # p-values of sampled SNPs (top hits + random) from the reference's arithmetic in float64
    out = chunkstore.open_container(str(tmp_path / "res.mmg"), "r")
    ps = np.concatenate([np.asarray(out["chrom_results"][c]["ps"][...]) for c in chroms])
    assert len(ps) == len(kept)
    lmm = lm.LinearMixedModel(y, ctx=ctx)
    lmm.add_random_effect(K)
    est = lmm.get_estimates(lmm._get_eigen_L_(), method="REML")
    assert abs(est["pseudo_heritability"] - res["pseudo_heritability"]) < 1e-7
    H = np.asarray(est["H_sqrt_inv"])
    hX, Yt = H @ lmm.X, H @ y
    r = Yt - hX @ np.linalg.lstsq(hX, Yt, rcond=None)[0]
    h0_rss = float(r @ r)
    Q, _ = np.linalg.qr(hX)
    sample = np.unique(np.r_[np.argsort(ps)[:6], rng.choice(len(kept), 14, replace=False)])
    worst = 0.0
    for gi in sample:
        t = H @ kept[gi].astype(np.float64)
        t = t - Q @ (Q.T @ t)
        F = (h0_rss / (h0_rss - float(t @ r) ** 2 / float(t @ t)) - 1) * (n - 2)
        p = float(orc.f_sf(np.array([F]), 1, n - 2)[0])
        if p > 1e-290:
            worst = max(worst, abs(ps[gi] / p - 1))
    assert worst < 1e-6


@pytest.mark.parametrize("n,m,adaptive,rows", [(150, 900, 0, True), (700, 2600, 4, True), (1100, 2000, 4, True),
                                                (240, 700, 4, True), (241, 700, 4, False), (256, 500, 0, False)])
def test_diploid_store_scan_takes_its_square_terms_from_the_bit_image(ctx, monkeypatch, n, m, adaptive, rows):
    """Stores of 0/1/2 codes (plink2hdf5.py:171-179), where s^2 != s.  The linear by-products of the scan GEMM (s.w, sum s,
    sum A_ii s_i) hold for any store; sum A_ii s_i^2 = sum A_ii s_i + 2 sum A_ii [s_i = 2] and sum s^2 = sum s + 2 #[s = 2]
    take their second terms from a bit image of the store (1/8 of its bytes; lin_hi_bits_kernel), built on the SECOND scan of
    the same content -- the first scan, and every scan when the model has no free rows for the linear terms (N mod 256 >
    240), is the finalize pass over the store's bytes, bit-identical to MMG_SCAN_FUSED_LINEAR=0.  Both against float64
    numpy; a write to the store invalidates the image."""
    rng = np.random.RandomState(1000 + n)
    f = rng.uniform(0.05, 0.95, size=(m, 1))
    snps = ((rng.random_sample((m, n)) < f).astype(np.int8) + (rng.random_sample((m, n)) < f).astype(np.int8))
    snps[3] = 0
    snps[4] = 2
    snps[5] = 1
    B = rng.standard_normal((n, 20)) / 4
    A = np.eye(n) * rng.uniform(0.5, 3.0, size=n) + B @ B.T / n
    A = 0.5 * (A + A.T)
    w = rng.standard_normal(n) * np.exp(rng.uniform(-6, 2, size=n))
    g = ctx.geno(snps)
    ctx.scan_set_model(A, w, adaptive)

    def check(out, S):
        scale = np.maximum(np.abs(S) @ np.abs(w), 1e-300)
        den = np.einsum("ij,ij->i", S @ A, S)
        assert np.max(np.abs(out["dot"] - S @ w) / scale) < 1e-13
        assert np.array_equal(out["sum"], S.sum(1))
        ok = den > 1e-6 * np.abs(den).max()
        assert rel(out["den"][ok], den[ok]) < 1e-7

    monkeypatch.setenv("MMG_SCAN_FUSED_LINEAR", "0")
    plain = ctx.scan(g, 5e6, n - 2, stats=True)
    monkeypatch.delenv("MMG_SCAN_FUSED_LINEAR")
    S = snps.astype(np.float64)
    check(plain, S)
    g2 = ctx.geno(snps)                                                      # a fresh store: scan count 0
    first = ctx.scan(g2, 5e6, n - 2, stats=True)
    for k in ("dot", "den", "ps", "rss"):
        assert np.array_equal(first[k], plain[k]), k                         # first scan: the same route
    second = ctx.scan(g2, 5e6, n - 2, stats=True)
    check(second, S)
    assert rel(second["den"], plain["den"]) < 1e-13
    keep = plain["ps"] > 1e-280
    assert rel(second["ps"][keep], plain["ps"][keep]) < 1e-11
    assert np.array_equal(second["rss"][3], plain["rss"][3])                 # monomorphic 0: rss = h0_rss either way
    same = all(np.array_equal(second[k], plain[k]) for k in ("dot", "den", "ps"))
    assert same != rows, "the image route was %staken" % ("not " if rows else "")
    third = ctx.scan(g2, 5e6, n - 2, stats=True)
    for k in ("dot", "den", "ps", "rss"):
        assert np.array_equal(third[k], second[k]), k                        # reproducible
    # a write invalidates the image: new rows, two more scans, both right
    snps2 = snps.copy()
    snps2[10:60] = snps[200:250]
    snps2[7] = 2 - snps[7]
    g2.upload(snps2[:64], 0)
    S2 = snps2.astype(np.float64)
    check(ctx.scan(g2, 5e6, n - 2, stats=True), S2)
    check(ctx.scan(g2, 5e6, n - 2, stats=True), S2)
    g.close()
    g2.close()
    if rows:                                                                 # MMG_SCAN_HI2=1: the image already on the first scan
        monkeypatch.setenv("MMG_SCAN_HI2", "1")
        g3 = ctx.geno(snps)
        forced = ctx.scan(g3, 5e6, n - 2, stats=True)
        monkeypatch.delenv("MMG_SCAN_HI2")
        g3.close()
        for k in ("dot", "den", "ps", "rss"):
            assert np.array_equal(forced[k], second[k]), k


def test_mlmm_on_a_store_of_diploid_codes_with_and_without_the_bit_image(ctx, monkeypatch):
    """MLMM scans ONE resident store once per step (linear_models.py:2543-2923): on 0/1/2 codes the first scan takes the
    finalize pass over the store, every later one the [s = 2] bit image.  Same step statistics, cofactors and p-values as
    with the image switched off (MMG_SCAN_HI2=0), and the first step's p-values against the oracle's emmax."""
    from mixmogam_amd import linear_models as lm, kinship
    from oracle import emmax_oracle as orc
    rng = np.random.RandomState(61)
    n, m = 300, 2500
    f = rng.uniform(0.1, 0.9, size=(m, 1))
    snps = ((rng.random_sample((m, n)) < f).astype(np.int8) + (rng.random_sample((m, n)) < f).astype(np.int8))
    snps = snps[(snps.std(1) > 0)]
    m = len(snps)
    K = orc.calc_ibd_kinship(snps)
    y = 0.9 * snps[100] - 0.7 * snps[900] + 0.5 * snps[1700] + rng.standard_normal(n)
    kw = dict(num_steps=4, forward_backwards=True, snps=snps, positions=list(range(m)), chromosomes=[1] * m, ctx=ctx,
              save_pvals=True)
    a = lm.mlmm(list(y), K, **kw)
    monkeypatch.setenv("MMG_SCAN_HI2", "0")
    b = lm.mlmm(list(y), K, **kw)
    monkeypatch.delenv("MMG_SCAN_HI2")
    assert len(a["step_info_list"]) == len(b["step_info_list"]) >= 5
    for sa, sb in zip(a["step_info_list"], b["step_info_list"]):
        assert [c[1] for c in sa["cofactors"]] == [c[1] for c in sb["cofactors"]]
        for k in ("pseudo_heritability", "bic", "e_bic", "m_bic", "min_pval", "rss", "mahalanobis_rss"):
            if sa[k] is None:
                assert sb[k] is None
            else:
                assert rel(np.asarray(sa[k], dtype=np.float64), np.asarray(sb[k], dtype=np.float64)) < 1e-9, k
        if "ps" in sa and sa["ps"] is not None:
            pa, pb = np.asarray(sa["ps"]), np.asarray(sb["ps"])
            ok = pb > 1e-280
            assert rel(pa[ok], pb[ok]) < 1e-9
    assert a["opt_dict"] == b["opt_dict"]
    first = np.asarray(a["step_info_list"][0]["ps"])
    want = orc.emmax(snps, y, K)["ps"]
    assert rel(first, want) < 1e-6

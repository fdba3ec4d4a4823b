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
    # cofactors as one array (rows = cofactors): the reference's `if cofactors:` raises on that; here it is the list again
    if case["cof"] is not None:
        as_rows = np.asarray([np.asarray(c, dtype=np.float64).reshape(-1) for c in case["cof"]])
        res2 = lm.emmax(list(case["snps"]), list(case["y"]), case["dbl_ibs_scaled"], cofactors=as_rows, ctx=ctx)
        assert np.array_equal(res2["ps"], res["ps"])
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
    stages = {}
    a = hdf5_data.run_emmax(src, y, min_maf=0.1, chunk_size=64, ctx=ctx, timings=stages)
    # what bench.py --mode c5 prints per rank: every stage of the pipeline, and the route the REML took
    assert {"kinship_pass_s", "reml_s", "scan_model_s", "scan_pass_s", "gather_s", "route"} <= set(stages)
    assert all(stages[k] >= 0.0 for k in stages if k.endswith("_s"))
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


# ---------------------------------------------------------------- containers
def test_chunkstore_mirrors_the_h5py_calls_the_reference_makes(tmp_path):
    from mixmogam_amd import chunkstore
    path = str(tmp_path / "geno.mmg")
    f = chunkstore.open_container(path)                                   # like h5py.File(name): created on demand
    gg = f.create_group("genot_data")
    cg = gg.create_group("chrom_2")
    snps = np.random.RandomState(0).randint(0, 2, size=(37, 11)).astype(np.int8)
    cg.create_dataset("raw_snps", compression="lzf", data=snps)           # plink2hdf5.py:111 call shape
    cg.create_dataset("freqs", data=snps.mean(1))
    gg.create_group("chrom_10")
    f.create_dataset("num_snps", data=np.array(37))
    stream = f.create_dataset("big", shape=(5, 3), dtype=np.int8)         # streamed writer
    stream[2:4] = 7
    f.flush(); f.close()
    g = chunkstore.open_container(path, "r")
    assert g.keys() == ["big", "genot_data", "num_snps"] and "kinship" not in g.keys()
    assert list(g["genot_data"].keys()) == ["chrom_10", "chrom_2"]      # name order, as h5py iterates
    raw = g["genot_data"]["chrom_2"]["raw_snps"]
    assert isinstance(raw, np.memmap) and len(raw) == 37
    assert np.array_equal(raw[5:9], snps[5:9]) and np.array_equal(raw[...], snps)
    assert int(g["num_snps"][...]) == 37 and np.array_equal(g["big"][...][2:4], np.full((2, 3), 7))
    with pytest.raises(IOError):
        g.create_dataset("x", data=np.zeros(2))
    a = chunkstore.open_container(path, "a")
    del a["big"]
    assert "big" not in a and "num_snps" in a
    with pytest.raises(ValueError):
        a.create_dataset("num_snps", data=np.array(1))                    # h5py refuses to overwrite too


def test_kinship_file_round_trip(tmp_path):
    k = np.random.RandomState(1).rand(6, 6)
    k = k + k.T
    accs = ["a%d" % i for i in range(6)]
    path = str(tmp_path / "k.mmg")
    kinship.save_kinship_to_file(path, k, accs, 1234)
    d = kinship.load_kinship_from_file(path, scaled=False)
    assert np.array_equal(d["k"], k) and d["accessions"] == accs and d["n_snps"] == 1234
    sub = kinship.load_kinship_from_file(path, accessions=["a4", "a1"], scaled=True)
    assert np.allclose(sub["k"], kinship.scale_k(k[[4, 1]][:, [4, 1]]))
    kinship.save_kinship_in_text_format(str(tmp_path / "k.csv"), k, accs)
    rows = open(str(tmp_path / "k.csv")).read().strip().split("\n")
    assert len(rows) == 6 and rows[2].split(",")[0] == "a2" and float(rows[2].split(",")[3]) == k[2, 2]


def test_run_emmax_from_and_to_containers(ctx, tmp_path):
    """hdf5_data.run_emmax / run_emmax_perm with the reference's call shape (file names in, result file out):
    same numbers as the in-memory mapping, datasets named as hdf5_data.py:146-184,241-347 names them."""
    from mixmogam_amd import chunkstore, simulations
    path = simulations.write_synthetic_container(str(tmp_path / "in.mmg"), 60, 700, chunk_rows=128, num_chroms=3,
                                                 num_causals=5)
    src = hdf5_data.open_hdf5(path)
    assert list(src["genot_data"].keys()) == ["chrom_1", "chrom_2", "chrom_3"] and len(src["phenotypes"]) == 60
    # the container holds exactly what synthetic_chunk regenerates (chunk ids run across chromosomes)
    assert np.array_equal(src["genot_data"]["chrom_1"]["raw_snps"][:128], simulations.synthetic_chunk(0, 128, 60))
    out_path = str(tmp_path / "res.mmg")
    res = hdf5_data.run_emmax(path, out_path, min_maf=0.1, chunk_size=100, ctx=ctx)
    mem = {c: {k: np.asarray(src["genot_data"][c][k][...]) for k in ("raw_snps", "freqs", "positions")}
           for c in src["genot_data"].keys()}
    ref = hdf5_data.run_emmax(mem, src["phenotypes"], min_maf=0.1, chunk_size=10 ** 6, ctx=ctx)
    o = chunkstore.open_container(out_path, "r")
    assert sorted(o.keys()) == ["chrom_results", "max_ll", "num_snps", "pseudo_heritability", "ve", "vg"]
    assert int(o["num_snps"][...]) == 700                                 # :150 copies the input file's count
    for c in mem:
        assert rel(o["chrom_results"][c]["ps"][...], ref["chrom_results"][c]["ps"]) < 1e-9
        assert np.array_equal(o["chrom_results"][c]["positions"][...], ref["chrom_results"][c]["positions"])
        assert np.array_equal(res["chrom_results"][c]["ps"], o["chrom_results"][c]["ps"][...])
    assert rel(o["pseudo_heritability"][...], ref["pseudo_heritability"]) < 1e-9
    # permutation variant: kinship + sorted minima + the reference's 5 % entries
    pidx = [np.random.RandomState(40 + p).permutation(60) for p in range(40)]
    outp = str(tmp_path / "perm.mmg")
    rp = hdf5_data.run_emmax_perm(path, outp, min_maf=0.1, chunk_size=100, num_perm=40, perm_idx=pidx, ctx=ctx)
    op = chunkstore.open_container(outp, "r")
    assert {"kinship", "perm_min_ps", "perm_max_f_stats", "five_perc_perm_min_ps",
            "five_perc_perm_max_f_stats"} <= set(op.keys())
    assert np.array_equal(op["perm_min_ps"][...], np.sort(rp["perm_min_ps"]))
    assert float(op["five_perc_perm_min_ps"][...]) == np.sort(rp["perm_min_ps"])[2]
    assert float(op["five_perc_perm_max_f_stats"][...]) == np.sort(rp["perm_max_f_stats"])[2]   # ascending (:341 quirk)
    assert rel(op["kinship"][...], ref["kinship"]) < 1e-9
    # stored kinship route: calculate_ibd_kinship writes 'kinship' into the input file, run_emmax reuses it
    k, n_snps = hdf5_data.calculate_ibd_kinship(path, chunk_size=90, ctx=ctx)
    assert n_snps == 700 and "kinship" in chunkstore.open_container(path, "r").keys()
    again = hdf5_data.run_emmax(path, None, min_maf=0.1, recalculate_kinship=False, chunk_size=100, ctx=ctx)
    assert rel(again["kinship"], k) == 0.0


def test_eigen_free_reml_and_scan_model_equal_the_eigen_route(case, ctx):
    """get_estimates_eigen_free / scan_model_eigen_free (likelihood sums and the scan model from H = K + delta I
    directly, no eig_L / eig_R) against the reference's numbers: same delta, ve, vg, max_ll, p-values."""
    lmm = lm.LinearMixedModel(list(case["y"]), ctx=ctx)
    lmm.add_random_effect(case["dbl_ibs_scaled"])
    if case["cof"] is not None:
        for c in case["cof"]:
            lmm.add_factor(c)
    res = lmm.get_estimates_eigen_free()
    assert res["H_sqrt_inv"] is None and res["n_factorisations"] >= 51
    for k in ("pseudo_heritability", "ve", "vg", "max_ll"):
        assert rel(res[k], case["dbl_emmax_" + k]) < 1e-7, k
    prep = lmm.scan_model_eigen_free(res)
    assert rel(prep["h0_rss"], case["dbl_emmax_h0_rss"]) < 1e-7
    assert rel(prep["h0_betas"], case["dbl_emmax_h0_betas"]) < 1e-6
    out = ctx.scan(ctx.geno(case["snps"]), prep["h0_rss"], prep["n_p"])
    assert rel(out["ps"], case["dbl_emmax_ps"]) < 1e-6


def test_run_emmax_over_a_never_resident_source(ctx):
    """hdf5_data.run_emmax over simulations.lazy_synthetic_source (rows regenerated on every read, config 5's "never
    fully resident") == the same run over the materialised arrays."""
    from mixmogam_amd import simulations
    tree, y = simulations.lazy_synthetic_source(50, 900, num_chroms=3, gen_rows=64, num_causals=5, threads=3)
    mem = {c: {"raw_snps": v["raw_snps"][:], "freqs": v["freqs"], "positions": v["positions"]} for c, v in tree.items()}
    a = hdf5_data.run_emmax(tree, y, min_maf=0.1, chunk_size=100, ctx=ctx)
    b = hdf5_data.run_emmax(mem, y, min_maf=0.1, chunk_size=10 ** 6, ctx=ctx)
    assert a["num_snps"] == b["num_snps"] == 900
    for c in mem:
        assert rel(a["chrom_results"][c]["ps"], b["chrom_results"][c]["ps"]) < 1e-9


def test_run_emmax_over_a_never_resident_packed_source(ctx):
    """The same with 1-bit rows that never exist expanded on the host (lazy_synthetic_source(packed=True): the
    generator's words are the `raw_snps_packed` rows) == the run over the expanded int8 arrays; N not a multiple of 8."""
    from mixmogam_amd import simulations, _lib
    tree, y = simulations.lazy_synthetic_source(53, 900, num_chroms=3, gen_rows=64, num_causals=5, threads=3, packed=True)
    ds = tree["chrom_2"]["raw_snps_packed"]
    assert ds.shape == (len(ds), 7) and ds[:].dtype == np.uint8 and np.array_equal(ds[70:200], ds[:][70:200])
    assert np.array_equal(ds[131], ds[:][131]) and not np.any(ds[:][:, -1] >> 5)          # pad bits are zero
    mem = {c: {"raw_snps": _lib.unpack_genotypes(v["raw_snps_packed"][:], 53, 1), "freqs": v["freqs"],
               "positions": v["positions"]} for c, v in tree.items()}
    assert abs(np.concatenate([m["raw_snps"] for m in mem.values()]).mean() - 0.5) < 0.01
    a = hdf5_data.run_emmax(tree, y, min_maf=0.1, chunk_size=100, ctx=ctx)
    b = hdf5_data.run_emmax(mem, y, min_maf=0.1, chunk_size=10 ** 6, ctx=ctx)
    assert a["num_snps"] == b["num_snps"] == 900
    for c in mem:
        assert rel(a["chrom_results"][c]["ps"], b["chrom_results"][c]["ps"]) < 1e-9


def test_run_emmax_over_a_never_resident_two_bit_source_of_diploid_codes(ctx):
    """lazy_synthetic_source(packed=2): 0/1/2 codes (plink2hdf5.py:171-179) as 2-bit rows, each the sum of two Bernoulli(0.5)
    alleles, never a 3; == the run over the expanded int8 arrays; N not a multiple of 4."""
    from mixmogam_amd import simulations, _lib
    tree, y = simulations.lazy_synthetic_source(53, 900, num_chroms=3, gen_rows=64, num_causals=5, threads=3, packed=2)
    ds = tree["chrom_2"]["raw_snps_packed"]
    assert int(tree["chrom_2"]["packed_bits"]) == 2
    assert ds.shape == (len(ds), 14) and ds[:].dtype == np.uint8 and np.array_equal(ds[70:200], ds[:][70:200])
    assert not np.any(ds[:][:, -1] >> 2)                                                   # 53 = 13 * 4 + 1: pad fields are zero
    mem = {c: {"raw_snps": _lib.unpack_genotypes(v["raw_snps_packed"][:], 53, 2), "freqs": v["freqs"],
               "positions": v["positions"]} for c, v in tree.items()}
    codes = np.concatenate([m["raw_snps"] for m in mem.values()])
    assert codes.max() == 2 and np.abs(np.bincount(codes.ravel(), minlength=3) / codes.size - [0.25, 0.5, 0.25]).max() < 0.01
    assert np.array_equal(_lib.pack_genotypes(mem["chrom_2"]["raw_snps"], 2), ds[:])
    a = hdf5_data.run_emmax(tree, y, min_maf=0.1, chunk_size=100, ctx=ctx)
    b = hdf5_data.run_emmax(mem, y, min_maf=0.1, chunk_size=10 ** 6, ctx=ctx)
    assert a["num_snps"] == b["num_snps"] == 900
    for c in mem:
        assert rel(a["chrom_results"][c]["ps"], b["chrom_results"][c]["ps"]) < 1e-9


def test_run_emmax_multi_equals_run_emmax_per_phenotype(ctx, tmp_path):
    """hdf5_data.run_emmax_multi (one pass over the chunks for all phenotypes) == run_emmax once per phenotype."""
    from mixmogam_amd import chunkstore, simulations
    path = simulations.write_synthetic_container(str(tmp_path / "in.mmg"), 60, 500, chunk_rows=128, num_chroms=2,
                                                 num_causals=5)
    src = hdf5_data.open_hdf5(path)
    rng = np.random.RandomState(3)
    raw1 = np.asarray(src["genot_data"]["chrom_1"]["raw_snps"][...])
    ys = np.vstack([src["phenotypes"], rng.randn(60) + 2 * raw1[7], rng.randn(60)])
    out = hdf5_data.run_emmax_multi(path, str(tmp_path / "multi.mmg"), phenotypes=ys, min_maf=0.1, chunk_size=100, ctx=ctx)
    o = chunkstore.open_container(str(tmp_path / "multi.mmg"), "r")
    for p in range(3):
        one = hdf5_data.run_emmax(path, None, phenotypes=ys[p], min_maf=0.1, chunk_size=100, k=out["kinship"], ctx=ctx)
        assert rel(out["pseudo_heritability"][p], one["pseudo_heritability"]) < 1e-9
        for c in one["chrom_results"]:
            assert rel(out["chrom_results"][c]["ps"][p], one["chrom_results"][c]["ps"]) < 1e-7
            assert np.array_equal(o["chrom_results"][c]["ps"][...][p], out["chrom_results"][c]["ps"][p])
    assert o["pseudo_heritability"][...].shape == (3,)


# ------------------------------------------------------------------ round 3
def test_return_transformed_snps_and_public_perm_test_host_logic(ctx):
    """_emmax_f_test_(return_transformed_snps=True) (:1309-1321,1355-1356) and emmax_perm_test / emmax_permutations
    (:1819-1841 / :1180-1230) through the numpy stand-in context against the reference's own outputs."""
    from conftest import load_extras2
    ex = load_extras2()
    n = int(ex["n"])
    for tag, cof in (("t", None), ("tc", [ex["cof"]])):
        lmm = lm.LinearMixedModel(list(ex["y"]), ctx=ctx)
        lmm.add_random_effect(ex["ibs_scaled"])
        if cof:
            lmm.add_factor(cof[0])
        r = lmm._emmax_f_test_(list(ex["snps"][:64]), ex["dbl_%s_H" % tag], return_transformed_snps=True, emma_num=0)
        assert isinstance(r["t_snps"], list) and len(r["t_snps"]) == 64 and r["t_snps"][0].shape == (n,)
        ref = ex["dbl_%s_snps" % tag]
        assert np.max(np.abs(np.asarray(r["t_snps"]) - ref)) < 1e-11 * np.max(np.abs(ref))
        assert rel(r["ps"], ex["dbl_%s_ps" % tag]) < 1e-7
    k = int(ex["perm_num_snps"])
    res = lm.emmax_perm_test(list(ex["snps"][:k]), list(ex["y"]), ex["ibs_scaled"], num_perm=len(ex["dbl_pub_perm_idx"]),
                             perm_idx=ex["dbl_pub_perm_idx"], H_sqrt_inv=ex["dbl_pub_perm_H"], reference_indexing=True,
                             ctx=ctx)
    assert rel(res["max_f_stats"], ex["dbl_pub_max_f_stats"]) < 1e-8
    assert rel(res["min_ps"], ex["dbl_pub_min_ps"]) < 1e-7
    p_f = sorted(zip(ex["dbl_pub_min_ps"], ex["dbl_pub_max_f_stats"]))
    assert rel(res["threshold_05"], p_f[len(p_f) // 20]) < 1e-7           # :1831
    with pytest.raises(IndexError):                                        # :1213 beyond num_perm SNPs
        lm.emmax_perm_test(list(ex["snps"][:40]), list(ex["y"]), ex["ibs_scaled"], num_perm=24,
                           perm_idx=ex["dbl_pub_perm_idx"], H_sqrt_inv=ex["dbl_pub_perm_H"], reference_indexing=True, ctx=ctx)
    # default reduction (per permutation, over the SNPs): same global optimum; H computed by the wrapper itself
    res2 = lm.emmax_perm_test(list(ex["snps"][:k]), list(ex["y"]), ex["ibs_scaled"], num_perm=24,
                              perm_idx=ex["dbl_pub_perm_idx"], H_sqrt_inv=ex["dbl_pub_perm_H"], ctx=ctx)
    assert abs(res2["max_f_stats"].max() / ex["dbl_pub_max_f_stats"].max() - 1) < 1e-8
    res3 = lm.emmax_perm_test(list(ex["snps"][:k]), list(ex["y"]), ex["ibs_scaled"], num_perm=5, ctx=ctx)
    assert len(res3["min_ps"]) == 5 and np.all(res3["min_ps"] <= 1)


def test_real_hdf5_files_drive_the_same_driver(ctx, tmp_path):
    """The real-HDF5 branch of chunkstore.open_container (the reference's own files: hdf5_data.py:77,146,199,241;
    kinship.py:149): an h5py.File with the layout of plink2hdf5.py:27-28,111-118,226 (lzf-compressed int8 raw_snps)
    goes through run_emmax / copy_tree / the kinship file helpers and gives the numbers of the directory container.
    Skipped where h5py is not installed (this image)."""
    h5py = pytest.importorskip("h5py")
    from mixmogam_amd import chunkstore, simulations
    dpath = simulations.write_synthetic_container(str(tmp_path / "in.mmg"), 60, 500, chunk_rows=128, num_chroms=2,
                                                  num_causals=5)
    hpath = str(tmp_path / "in.hdf5")
    with h5py.File(hpath, "w") as h5f:
        d = chunkstore.open_container(dpath, "r")
        gg = h5f.create_group("genot_data")
        for c in d["genot_data"].keys():
            cg = gg.create_group(c)
            cg.create_dataset("raw_snps", data=np.asarray(d["genot_data"][c]["raw_snps"][...]), compression="lzf",
                              chunks=(64, 60))
            for k in ("positions", "freqs"):
                cg.create_dataset(k, data=np.asarray(d["genot_data"][c][k][...]))
        ig = h5f.create_group("indiv_data")
        for k in d["indiv_data"].keys():
            ig.create_dataset(k, data=np.asarray(d["indiv_data"][k][...]))
        h5f.create_dataset("num_snps", data=np.asarray(d["num_snps"][...]))
    f = chunkstore.open_container(hpath, "r")
    assert isinstance(f, h5py.File)
    f.close()
    ref = hdf5_data.run_emmax(dpath, str(tmp_path / "res.mmg"), min_maf=0.1, chunk_size=100, ctx=ctx)
    res = hdf5_data.run_emmax(hpath, str(tmp_path / "res.hdf5"), min_maf=0.1, chunk_size=100, ctx=ctx)
    with h5py.File(str(tmp_path / "res.hdf5"), "r") as o:
        assert sorted(o.keys()) == ["chrom_results", "max_ll", "num_snps", "pseudo_heritability", "ve", "vg"]
        for c in ref["chrom_results"]:
            assert np.array_equal(o["chrom_results"][c]["ps"][...], ref["chrom_results"][c]["ps"])
            assert np.array_equal(res["chrom_results"][c]["ps"], ref["chrom_results"][c]["ps"])
    # HDF5 -> directory container and back
    back = chunkstore.Store(str(tmp_path / "copy.mmg"), "w")
    with h5py.File(hpath, "r") as h5f:
        chunkstore.copy_tree(h5f, back)
    for c in ref["chrom_results"]:
        assert np.array_equal(back["genot_data"][c]["raw_snps"][...],
                              chunkstore.open_container(dpath, "r")["genot_data"][c]["raw_snps"][...])
    kpath = str(tmp_path / "k.hdf5")
    kinship.save_kinship_to_file(kpath, ref["kinship"], [str(i) for i in range(60)], n_snps=500)
    with h5py.File(kpath, "r") as kf:
        assert sorted(kf.keys()) == ["accessions", "kinship", "n_snps"]        # kinship.py:163-166
    kd = kinship.load_kinship_from_file(kpath, scaled=False)
    assert np.array_equal(kd["k"], ref["kinship"]) and kd["n_snps"] == 500


def test_real_hdf5_branch_runs_under_an_interpreter_that_has_h5py():
    """This image's main interpreter has no h5py (the test above is skipped there), but it ships a second one that has
    (/opt/conda/bin/python3.9: h5py 3.3.0, numpy 1.26, scipy 1.7): the real-HDF5 branch -- h5py.File in, h5py.File out,
    lzf-compressed raw_snps in the layout of plink2hdf5.py:111-118, copy_tree, the kinship file helpers -- runs there, on the
    numpy stand-in context.  Skipped where neither interpreter has h5py."""
    import importlib.util, os, subprocess, sys
    if importlib.util.find_spec("h5py") is not None:
        pytest.skip("h5py is importable here: test_real_hdf5_files_drive_the_same_driver ran in this interpreter")
    alt = os.environ.get("MMG_H5PY_PYTHON", "/opt/conda/bin/python3.9")
    probe = subprocess.run([alt, "-c", "import h5py, numpy, scipy, pytest"], capture_output=True) if os.path.exists(alt) else None
    if probe is None or probe.returncode != 0:
        pytest.skip("no interpreter with h5py on this machine")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([alt, "-m", "pytest", os.path.join(root, "tests", "test_host_logic.py"), "-q", "-p", "no:cacheprovider",
                          "-k", "real_hdf5_files_drive", "-W", "ignore"], capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0 and "1 passed" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


@pytest.mark.parametrize("bits,hi", [(1, 2), (2, 3)])
def test_packed_genotype_containers_drive_the_same_driver(ctx, tmp_path, bits, hi):
    """`raw_snps_packed` (1 / 2 bits per genotype, low bits first) instead of `raw_snps`: pack / unpack round trip
    and run_emmax over the packed container == run_emmax over the int8 one (host logic; the device unpack is
    tests/test_gpu_round3.py)."""
    from mixmogam_amd import _lib, chunkstore
    rng = np.random.RandomState(bits)
    n = 61                                                                # not a multiple of 8 or 4: ragged last byte
    snps = {"chrom_%d" % c: rng.randint(0, hi, size=(150 + 10 * c, n)).astype(np.int8) for c in (1, 2)}
    for c in snps:
        p = _lib.pack_genotypes(snps[c], bits)
        assert p.dtype == np.uint8 and p.shape == (len(snps[c]), (n * bits + 7) // 8)
        assert np.array_equal(_lib.unpack_genotypes(p, n, bits), snps[c])
    assert np.array_equal(_lib.pack_genotypes(np.array([[1, 0, 0, 0, 0, 0, 0, 0, 1]]), 1), [[1, 1]])        # LSB first
    assert np.array_equal(_lib.pack_genotypes(np.array([[1, 2, 3, 0, 2]]), 2), [[1 | 2 << 2 | 3 << 4, 2]])  # .bed order
    with pytest.raises(ValueError):
        _lib.pack_genotypes(np.array([[0, 2]]), 1)
    y = rng.randn(n) + snps["chrom_1"][7]
    a = chunkstore.write_genotype_container(str(tmp_path / "plain.mmg"), snps, range(n), phenotypes=y)
    b = chunkstore.write_genotype_container(str(tmp_path / "packed.mmg"), snps, range(n), phenotypes=y, packed_bits=bits)
    g = chunkstore.open_container(b, "r")["genot_data"]["chrom_1"]
    assert "raw_snps" not in g and int(g["packed_bits"][...]) == bits and int(g["num_indivs"][...]) == n
    ra = hdf5_data.run_emmax(a, None, min_maf=0.05, chunk_size=64, ctx=ctx)
    rb = hdf5_data.run_emmax(b, None, min_maf=0.05, chunk_size=64, ctx=ctx)
    assert ra["num_snps"] == rb["num_snps"]
    for c in ra["chrom_results"]:
        assert np.array_equal(ra["chrom_results"][c]["ps"], rb["chrom_results"][c]["ps"])
    assert np.array_equal(ra["kinship"], rb["kinship"])


def test_chunk_plan_ramp_covers_every_row_once():
    """hdf5_data._chunk_plan(ramp=True): short first chunks (chunk/8, /4, /2) then full ones -- every kept row exactly
    once, in order, per chromosome."""
    rng = np.random.RandomState(0)
    src = {"c%d" % c: {"raw_snps": np.zeros((m, 7), dtype=np.int8), "freqs": rng.uniform(0.05, 0.5, size=m),
                       "positions": np.arange(m) * 3} for c, m in ((1, 5000), (2, 300), (3, 2600))}
    for ramp in (False, True):
        plan = hdf5_data._chunk_plan(src, 0.1, 2048, ramp=ramp)
        for chrom in src:
            keep = np.nonzero(np.minimum(src[chrom]["freqs"], 1 - src[chrom]["freqs"]) > 0.1)[0]
            got = np.concatenate([sel for c, sel, _p in plan if c == chrom])
            pos = np.concatenate([p for c, _s, p in plan if c == chrom])
            assert np.array_equal(got, keep) and np.array_equal(pos, keep * 3)
        sizes = [len(sel) for _c, sel, _p in plan]
        assert max(sizes) <= 2048
        if ramp:
            assert sizes[:3] == [256, 512, 1024]


@pytest.mark.parametrize("bits", [0, 1])
def test_read_chunk_into_a_bounded_staging_buffer(tmp_path, bits):
    """hdf5_data._read_chunk with a reusable staging buffer: a selection whose SPAN of file rows fits is read in one piece and
    compacted in place; a sparser one (a MAF filter that keeps few rows: span > buffer) is read run by run straight to its
    place; one that does not fit at all takes the mapping.  The buffer never has to hold more than the kept rows (advisor r4:
    it was sized by the span, 2 x 60 GB of page-locked memory at N = 50,000 with 10 % of the SNPs kept)."""
    from mixmogam_amd import chunkstore, _lib
    rng = np.random.RandomState(3)
    n, m = 37, 4000
    snps = (rng.random_sample((m, n)) < 0.4).astype(np.int8)
    path = str(tmp_path / "g")
    chunkstore.write_genotype_container(path, {"c1": snps}, np.arange(n), packed_bits=bits)
    gd = chunkstore.open_container(path, "r")["genot_data"]
    rows = _lib.pack_genotypes(snps, 1) if bits else snps
    row_bytes = rows.shape[1]
    for keep_frac, buf_rows in ((1.0, 600), (0.5, 600), (0.05, 60), (0.05, 10)):
        sel = np.sort(rng.choice(np.arange(100, 1100), size=max(2, int(1000 * keep_frac)), replace=False))
        if len(sel) > 500 and keep_frac == 1.0:
            sel = np.arange(100, 600)
        out = np.full(buf_rows * row_bytes, 0x55, dtype=np.int8)
        got = hdf5_data._read_chunk(gd, "c1", sel, out=out)
        assert np.array_equal(np.asarray(got).view(rows.dtype), rows[sel]), (keep_frac, buf_rows)
        fits = buf_rows >= len(sel)
        assert np.shares_memory(got, out) == fits                      # staged when the kept rows fit, else through the mapping


def test_merge_plan_joins_neighbouring_chunks_of_a_chromosome():
    """hdf5_data._merge_plan: the kinship pass groups the plan's chunks up to >= 65,536 SNPs per exact-GRM call (within a
    byte budget, never across chromosomes); rows and positions keep their order."""
    tree = {"c1": {"raw_snps": np.zeros((2500, 8), np.int8), "freqs": np.full(2500, 0.5), "positions": np.arange(2500)},
            "c2": {"raw_snps": np.zeros((700, 8), np.int8), "freqs": np.full(700, 0.5), "positions": np.arange(700)}}
    plan = hdf5_data._chunk_plan(tree, 0.1, 400)
    assert [len(s) for _c, s, _p in plan] == [400] * 6 + [100] + [400, 300]
    merged = hdf5_data._merge_plan(plan, 8, min_rows=1000)
    assert [(c, len(s)) for c, s, _p in merged] == [("c1", 1200), ("c1", 1200), ("c1", 100), ("c2", 700)]
    assert np.array_equal(np.concatenate([s for c, s, _p in merged if c == "c1"]), np.arange(2500))
    assert np.array_equal(np.concatenate([p for c, _s, p in merged if c == "c2"]), np.arange(700))
    capped = hdf5_data._merge_plan(plan, 8, min_rows=1000, max_bytes=8 * 900)     # at most 900 rows per chunk
    assert max(len(s) for _c, s, _p in capped) <= 900 and sum(len(s) for _c, s, _p in capped) == 3200
    assert hdf5_data._merge_plan(plan, 8, min_rows=1) == plan


def test_band_route_sums_are_not_dealt_over_ranks():
    """_SpectralSumsChol: with the band route every rank evaluates all deltas itself (one reduction of K serves them all);
    only the one-factorisation-per-delta route deals the grid out and gathers."""
    class Coll(object):
        rank, world = 1, 4
        def allgather(self, x):
            raise AssertionError("no exchange on the band route")
    class Reml(object):
        N = 1000
        calls = []
        def uses_band(self, route="auto"):
            return route != "chol"
        def sums(self, deltas, route="auto"):
            self.calls.append((len(deltas), route))
            d = np.asarray(deltas, dtype=np.float64)
            return d, 2 * d, 3 * d, 4 * d, 7.0
    r = Reml()
    s = lm._SpectralSumsChol(r, Coll())
    out = s.at(np.arange(1.0, 9.0))
    assert np.array_equal(out[1], 2 * np.arange(1.0, 9.0)) and s.sum_sq_etas == 7.0 and r.calls == [(8, "auto")]
    assert s.n_factorisations == 8
    with pytest.raises(AssertionError, match="no exchange"):
        lm._SpectralSumsChol(r, Coll(), route="chol").at(np.arange(1.0, 9.0))      # this one does go through the collective


def test_reml_search_on_the_interpolated_sums_matches_the_exact_search():
    """_SpectralSumsChol.prepare_interval (band route): the secant search of get_estimates (linear_models.py:847) runs on a
    16-node Chebyshev model of the four likelihood sums over the bracket, so a whole search costs TWO device calls
    (grid, nodes; the likelihood at the optimum comes from the model too) instead of one per secant step.  With a stand-in workspace whose sums are exact functions of a
    spectrum: the variance ratio agrees with the search on exact evaluations to 1e-11 relative, the likelihood at the
    optimum is an exact evaluation, and the model itself is good to 1e-12 anywhere in its interval."""
    rng = np.random.RandomState(3)
    n = 400
    pop = rng.randint(0, 3, size=n)
    f = np.clip(0.5 + 0.2 * rng.standard_normal((1500, 3)), 0.05, 0.95)
    S = (rng.random_sample((1500, n)) < f[:, pop]).astype(np.float64)
    S = S[S.std(1) > 0]
    Z = (S - S.mean(1, keepdims=True)) / S.std(1, keepdims=True)
    K = Z.T @ Z / len(Z)
    X = np.column_stack([np.ones(n), rng.standard_normal(n)])
    y = 1.5 * (K @ rng.standard_normal(n)) / np.sqrt(n) + rng.standard_normal(n)
    lam, U = np.linalg.eigh(K)
    exact = lm._SpectralSumsL({'values': lam, 'vectors': U.T}, X, y)

    class Reml(object):
        N = n
        calls = []
        def uses_band(self, route="auto"):
            return True
        def sums(self, deltas, route="auto"):
            self.calls.append(len(deltas))
            return tuple(exact.at(np.asarray(deltas, dtype=np.float64))) + (exact.sum_sq_etas,)

    model = lm.LinearMixedModel(list(y))
    model.add_factor(X[:, 1])

    def two_calls(reml):                                                            # round 4's scheme: the grid, then 16 Chebyshev nodes
        sm = lm._SpectralSumsChol(reml)
        sm.FINE_GRID = False
        return sm

    r1 = Reml(); r1.calls = []
    a = model.get_estimates(None, method='REML', _sums=two_calls(r1))
    assert r1.calls == [51, lm._SpectralSumsChol.INTERP_NODES], r1.calls           # grid, nodes (the optimum: from the model)
    assert a['n_device_calls'] == 2
    r3 = Reml(); r3.calls = []
    exact_final = two_calls(r3)
    exact_final.FINAL_FROM_MODEL = False                                           # the likelihood at the optimum from the workspace
    c = model.get_estimates(None, method='REML', _sums=exact_final)
    assert r3.calls == [51, lm._SpectralSumsChol.INTERP_NODES, 1]
    assert abs(a['max_ll'] - c['max_ll']) <= 1e-11 * abs(c['max_ll']) and abs(a['vg'] / c['vg'] - 1) < 1e-11
    r2 = Reml(); r2.calls = []
    plain = two_calls(r2)
    plain.prepare_interval = lambda lo, hi: None                                    # every secant step asks the workspace
    b = model.get_estimates(None, method='REML', _sums=plain)
    assert len(r2.calls) > 4 and 1e-3 < b['delta'] < 1e3                            # an interior optimum, several steps
    assert abs(a['delta'] / b['delta'] - 1) < 1e-11, (a['delta'], b['delta'])
    for k in ('max_ll', 've', 'vg', 'pseudo_heritability'):
        assert abs(a[k] - b[k]) <= 1e-10 * max(1.0, abs(b[k])), (k, a[k], b[k])
    # the model against exact evaluations across its interval
    s = two_calls(Reml())
    s.prepare_interval(0.5 * b['delta'], 1.5 * b['delta'])
    for d in np.exp(np.linspace(s._interp[0], s._interp[1], 37)):
        want = exact.at(np.array([d]))
        got = s.at(np.array([d]))
        for i in range(4):
            assert abs(got[i][0] - want[i][0]) <= 1e-12 * max(1.0, abs(want[i][0])), (d, i)
    # ---- round 5 (the default): the grid refined to a spacing of 0.1 with 13 nodes beyond either end, the search on the
    # polynomial through the 20 nodes around each question.  Without mmg_reml_band_factor: every refined node in ONE call
    assert lm._SpectralSumsChol.FINE_GRID
    n_fine = 50 * 4 + 1 + 2 * lm._SpectralSumsChol.FINE_PAD
    r4 = Reml(); r4.calls = []
    one = lm._SpectralSumsChol(r4)
    e = model.get_estimates(None, method='REML', _sums=one)
    assert r4.calls == [n_fine] and e['n_device_calls'] == 1, r4.calls
    # ... with it: all refined nodes factored in one sweep, sums on the grid, then on the refined nodes around the bracket
    class RemlKeep(Reml):
        def band_factor(self, deltas):
            self.factored = np.array(deltas, dtype=np.float64)
    r6 = RemlKeep(); r6.calls = []
    e6 = model.get_estimates(None, method='REML', _sums=lm._SpectralSumsChol(r6))
    assert len(r6.factored) == n_fine and r6.calls[0] == 51 and len(r6.calls) == 2 and 20 < r6.calls[1] < 60, r6.calls
    assert e6['n_factorisations'] == n_fine + r6.calls[1] and e6['n_device_calls'] == 2
    assert e6['delta'] == e['delta'] and e6['max_ll'] == e['max_ll'] and e6['vg'] == e['vg']     # the same nodes, the same polynomial
    assert abs(e['delta'] / b['delta'] - 1) < 1e-11, (e['delta'], b['delta'])
    for k in ('max_ll', 've', 'vg', 'pseudo_heritability'):
        assert abs(e[k] - b[k]) <= 1e-10 * max(1.0, abs(b[k])), (k, e[k], b[k])
    worst, worst_low = 0.0, 0.0
    for d in np.exp(np.linspace(-10.3, 10.3, 997)):                                 # anywhere a bracket (+- its margin) can lie
        dq = d * (1 + 1e-9)                                                         # (not a remembered node)
        one._interp = (np.log(dq) - 1e-9, np.log(dq) + 1e-9, None, None, None)
        want = exact.at(np.array([dq]))
        got = one.at(np.array([dq]))
        for i in range(4):
            err = abs(got[i][0] - want[i][0]) / max(1.0, abs(want[i][0]))
            if d > np.exp(-5.0):
                worst = max(worst, err)
            else:
                worst_low = max(worst_low, err)
    # (this K is singular: under delta = e^-5 the stand-in's own sums carry the rounding of a 1e5-conditioned H, and the
    # polynomial through 20 noisy values repeats it; stencils of 16 / 24 nodes and half the spacing give the same figures)
    assert worst <= 1e-12 and worst_low <= 1e-9, (worst, worst_low)
    # the grid of mlmm (:1574: 100 intervals of 0.2): refined twice -- the same spacing, one call
    r5 = Reml(); r5.calls = []
    f = model.get_estimates(None, method='REML', ngrids=100, _sums=lm._SpectralSumsChol(r5))
    assert r5.calls == [100 * 2 + 1 + 2 * lm._SpectralSumsChol.FINE_PAD]
    g2 = model.get_estimates(None, method='REML', ngrids=100, _sums=two_calls(Reml()))
    assert abs(f['delta'] / g2['delta'] - 1) < 1e-11 and abs(f['max_ll'] - g2['max_ll']) <= 1e-11 * abs(g2['max_ll'])


# ---------------------------------------------------------------------- round 4: surface closures on the numpy stand-in
def test_round4_surface_closures_against_the_reference_numbers():
    """fast_f_test(with_betas / Z), t_snps under with_betas, emmax_multi with four cofactors, ML without eigh and IBD
    kinship from pre-normalised `snps` datasets -- host logic through the numpy stand-in context (the same cases run on
    the device in tests/test_gpu_round4.py)."""
    import _round4_cases as r4
    from conftest import load_extras, load_extras3
    from fake_ctx import FakeContext
    ex3, ex = load_extras3(), load_extras()
    r4.fast_f_test_with_betas(FakeContext(), ex3, tol=1e-8)
    r4.transformed_snps_with_betas(FakeContext(), ex3, tol_t=1e-10, tol=1e-7)
    r4.emmax_multi_four_cofactors(FakeContext(), ex3, tol=1e-7)
    r4.ml_without_an_eigendecomposition(FakeContext(), ex, tol=1e-7)
    r4.ibd_kinship_from_normalised_snps(FakeContext())


def test_random_shapes_through_the_host_mirror():
    """tools/random_parity.py on the numpy stand-ins of the C ABI (tests/fake_ctx.py): the host side of kinship, emmax(),
    linear_model() and emmax_multi() -- dispatch by size, REML search, cofactors, genotype alphabets -- on 40 random small
    problems against the oracle.  (The GPU suite runs the same sweep on the device.)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "random_parity.py"), "40", "3"], capture_output=True, text=True,
                         timeout=600, env=dict(os.environ, MMG_PARITY_HOST="1"))
    assert out.returncode == 0 and "failures: 0" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]


# ------------------------------------------------------------------ round 6: the HDF5 drivers' HOST logic vs the reference's own runs
@pytest.mark.parametrize("variant", ["bin", "dip"])
def test_hdf5_driver_host_logic_vs_the_reference_run(tmp_path, variant):
    """The bodies of the -m gpu driver tests (tests/test_gpu_round6.py) with the numpy stand-in context: what is checked
    here is the drivers' own reading of hdf5_data.py (MAF filter, chromosome order, `chr12_snps`, sorted outputs, the
    5 % index, both num_snps conventions, file layout) against tests/golden/hdf5_n200.npz -- the reference's own files."""
    import test_gpu_round6 as g
    from conftest import load_hdf5_golden
    from fake_ctx import FakeContext
    d, ctx = load_hdf5_golden(), FakeContext()
    for i, (fn, args) in enumerate(((g.test_calculate_ibd_kinship_driver_vs_the_reference_run, (variant,)),
                                    (g.test_run_emmax_driver_vs_the_reference_run, (variant, 0)),
                                    (g.test_run_emmax_driver_vs_the_reference_run, (variant, 1 if variant == "bin" else 2)),
                                    (g.test_run_emmax_with_the_stored_kinship_vs_the_reference_run, (variant,)),
                                    (g.test_run_emmax_perm_driver_vs_the_reference_run, (variant,)))):
        sub = tmp_path / ("case%d" % i)
        sub.mkdir()
        fn(ctx, sub, d, *args)

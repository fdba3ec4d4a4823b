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

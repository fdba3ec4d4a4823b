import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


_BIG = {}


def load_case(name):
    if name == "struct_n1000_s7":
        return dict(load_n1000())
    if name == "ft10_config1":
        return dict(load_config1())
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    n = int(d["n"])
    d["snps"] = np.unpackbits(d["snps_packed"], axis=1)[:, :n].astype(np.int8)
    d["cof"] = list(d["cofactors"]) if len(d["cofactors"]) else None
    return d


CASE_NAMES = ["struct_n150_s0", "struct_n150_s1", "struct_n300_s2", "struct_n300_s3", "bern_n200_s4"]
# round 6: reference runs at a size that reaches the device code's tile structure (N = 1000: four tile rows, the tail launch,
# >= 15 band-reduction panels) and in the shape of BASELINE config 1 (the 198 FT10 accessions through the reference's own
# coordinate_w_phenotype_data); they carry the emmax / REML / permutation keys of the small cases, not the kernel-level ones
BIG_CASE_NAMES = ["struct_n1000_s7", "ft10_config1"]


@pytest.fixture(params=CASE_NAMES)
def case(request):
    return load_case(request.param)


@pytest.fixture(params=CASE_NAMES + BIG_CASE_NAMES)
def case_emmax(request):
    return load_case(request.param)


@pytest.fixture(params=CASE_NAMES + BIG_CASE_NAMES[:1])
def case_reml(request):
    return load_case(request.param)


def load_n1000():
    """tests/golden/struct_n1000_s7.npz with the keys of load_case.  The 8 MB matrices are rebuilt, then VERIFIED against what
    the reference run recorded: K = counts / 2M + 0.5 scaled (exact function of the genotypes; products K v and the diagonal
    are in the fixture), H_sqrt_inv of the permutation run from its row signs (reference_row_signs checks products H v)."""
    if "n1000" in _BIG:
        return _BIG["n1000"]
    from oracle import emmax_oracle as orc
    d = dict(np.load(os.path.join(GOLDEN, "struct_n1000_s7.npz")))
    n = int(d["n"])
    d["snps"] = np.unpackbits(d["snps_packed"], axis=1)[:, :n].astype(np.int8)
    d["cof"] = list(d["cofactors"])
    K = orc.calc_ibs_kinship(d["snps"])
    want = d["ibs_scaled_probe"]
    assert np.abs(K @ np.random.RandomState(98).randn(n, 3) - want).max() < 1e-12 * np.abs(want).max()
    assert np.abs(np.diag(K) - d["ibs_scaled_diag"]).max() < 1e-13
    d["dbl_ibs_scaled"] = K
    est0 = orc.get_estimates(d["y"], np.ones((n, 1)), orc.scale_k(K))
    assert abs(est0["delta"] / float(d["dbl_reml0_delta"]) - 1) < 1e-6
    d["dbl_perm_H"] = reference_row_signs(d, "dbl", est0["H_sqrt_inv"])
    d["dbl_perm_idx"] = d["lit_perm_idx"] = d["perm_idx"].astype(np.int64)
    _BIG["n1000"] = d
    return d


def load_config1():
    """tests/golden/ft10_config1.npz: `snps`, `y`, K are the COORDINATED data the reference handed to calc_ibs_kinship / emmax
    (examples.py:84-90); the raw inputs (`raw_snps`, accessions, chromosomes, positions) are what the plumbing test feeds the
    build's own snpsdata / phenotypeData."""
    if "config1" in _BIG:
        return _BIG["config1"]
    d = dict(np.load(os.path.join(GOLDEN, "ft10_config1.npz")))
    d["raw_snps"] = np.unpackbits(d["snps_packed"], axis=1)[:, :int(d["n_genotyped"])].astype(np.int8)
    n = len(d["coord_values"])
    d["n"] = np.int64(n)
    d["snps"] = np.unpackbits(d["coord_snps_packed"], axis=1)[:, :n].astype(np.int8)
    d["y"] = d["coord_values"]
    d["cof"] = None
    K = np.zeros((n, n))
    K[np.triu_indices(n)] = d["ibs_scaled_triu"]
    d["dbl_ibs_scaled"] = K + np.triu(K, 1).T
    _BIG["config1"] = d
    return d


def load_extras():
    """tests/golden/extras_n150.npz: replicates (Z), emma(), get_ML and the multi-phenotype loop of the reference."""
    d = dict(np.load(os.path.join(GOLDEN, "extras_n150.npz")))
    n = int(d["n"])
    d["snps"] = np.unpackbits(d["snps_packed"], axis=1)[:, :n].astype(np.int8)
    return d


def load_extras2():
    """tests/golden/extras2_n150.npz (round 3): t_snps of _emmax_f_test_(return_transformed_snps=True) and the public
    permutation test LinearMixedModel.emmax_permutations, both from the reference itself."""
    d = dict(np.load(os.path.join(GOLDEN, "extras2_n150.npz")))
    n = int(d["n"])
    d["snps"] = np.unpackbits(d["snps_packed"], axis=1)[:, :n].astype(np.int8)
    return d


def load_extras3():
    """tests/golden/extras3_n150.npz (round 4): LinearModel.fast_f_test(with_betas=True), _emmax_f_test_(with_betas=True,
    return_transformed_snps=True) and the reference's loop of emmax() with four cofactors, all from the reference itself."""
    d = dict(np.load(os.path.join(GOLDEN, "extras3_n150.npz")))
    n = int(d["n"])
    d["snps"] = np.unpackbits(d["snps_packed"], axis=1)[:, :n].astype(np.int8)
    return d


def load_hdf5_golden():
    """tests/golden/hdf5_n200.npz (round 6): what the reference's own hdf5_data.calculate_ibd_kinship / run_emmax /
    run_emmax_perm wrote for two 3-chromosome files (0/1 codes 'bin', 0/1/2 codes 'dip'), 'lit' and 'dbl' modes.
    Adds d[variant + '_chroms'] = [(name, raw_snps int8, freqs, positions)] and full symmetric kinship matrices."""
    d = dict(np.load(os.path.join(GOLDEN, "hdf5_n200.npz")))
    n = int(d["n"])
    iu = np.triu_indices(n)
    for variant in ("bin", "dip"):
        chroms = []
        for c in (1, 2, 3):
            name = "chrom_%d" % c
            s = np.unpackbits(d["%s_%s_raw_snps_ge1" % (variant, name)], axis=1)[:, :n].astype(np.int8)
            if variant == "dip":
                s = s + np.unpackbits(d["%s_%s_raw_snps_eq2" % (variant, name)], axis=1)[:, :n].astype(np.int8)
            chroms.append((name, s, d["%s_%s_freqs" % (variant, name)], d["%s_%s_positions" % (variant, name)]))
        d[variant + "_chroms"] = chroms
        for mode in ("lit", "dbl"):
            for key in ("perm_kinship", "calc_kinship") + (("calcnorm_kinship",) if variant == "bin" else ()):
                full = np.zeros((n, n))
                full[iu] = d["%s_%s_%s" % (variant, mode, key)]
                d["%s_%s_%s" % (variant, mode, key)] = full + np.triu(full, 1).T
    return d


def reference_row_signs(d, tag, H):
    """H (an H_sqrt_inv of the same K and delta, any LAPACK's row signs) with the row signs the REFERENCE run had
    (fixture keys <tag>_perm_H_argmax / _sign), verified against the recorded products H v (<tag>_perm_H_probe)."""
    H = np.array(H, dtype=np.float64)
    n = len(H)
    at = H[np.arange(n), d[tag + "_perm_H_argmax"]]
    H *= (np.sign(at) * d[tag + "_perm_H_sign"])[:, None]
    probe = H @ np.random.RandomState(99).randn(n, 3)
    want = d[tag + "_perm_H_probe"]
    assert np.abs(probe - want).max() < 1e-7 * np.abs(want).max(), "H_sqrt_inv does not reconstruct the reference's"
    return H

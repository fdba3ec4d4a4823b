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


def load_case(name):
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    n = int(d["n"])
    d["snps"] = np.unpackbits(d["snps_packed"], axis=1)[:, :n].astype(np.int8)
    d["cof"] = list(d["cofactors"]) if len(d["cofactors"]) else None
    return d


CASE_NAMES = ["struct_n150_s0", "struct_n150_s1", "struct_n300_s2", "struct_n300_s3", "bern_n200_s4"]


@pytest.fixture(params=CASE_NAMES)
def case(request):
    return load_case(request.param)


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
            for key in ("perm_kinship", "calc_kinship"):
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

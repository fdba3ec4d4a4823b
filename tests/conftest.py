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

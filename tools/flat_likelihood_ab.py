#!/usr/bin/env python3
"""Flat REML likelihoods (pure-noise phenotypes, h2 ~ 0): how far do the p-values of emmax() sit from the float64 oracle
by the eigen-free route (band / Cholesky sums + interpolant, the default) and by the eigen route (spectral sums, the reference's
own form)?  Round 5 asked this before adding a rule "flat likelihood -> eigen route": the answer (profiles/r5_flat_likelihood_ab.txt)
is that the two routes sit equally close to the oracle -- 1e-8 on p, 1e-13 on the pseudo-heritability -- also at h2 ~ 0, so no
such rule exists; where delta itself is ill-determined it is so for spectral sums as well.  Checker only.
    python tools/flat_likelihood_ab.py [cases] [seed]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mixmogam_amd import _lib, kinship, linear_models as lm
from oracle import emmax_oracle as orc
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = _lib.get_context()
rng = np.random.RandomState(seed)


def rel(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


worst = [0.0] * 4
print("%5s %5s %9s | %10s %10s | %10s %10s" % ("n", "m", "ref h2", "free p", "free h2", "eigen p", "eigen h2"))
for c in range(cases):
    n = int(rng.choice([40, 120, 300, 700, 1500]))
    m = int(rng.randint(200, 1500))
    freq = rng.uniform(0.05, 0.95, m)
    snps = (rng.random_sample((m, n)) < freq[:, None]).astype(np.int8)
    snps = snps[snps.std(1) > 0]
    effect = float(rng.choice([0.0, 0.0, 0.1, 0.3]))
    y = rng.standard_normal(n) + effect * snps[rng.randint(len(snps))]
    K = kinship.calc_ibs_kinship(snps, ctx=ctx)
    ref = orc.emmax(snps, y, K)
    out = []
    for min_n in (15, 10 ** 9):                                       # eigen-free; eigen route
        lm.EIGEN_FREE_MIN_N = min_n
        res = lm.emmax(snps, list(y), K, emma_num=0, ctx=ctx)
        out += [rel(res["ps"], ref["ps"]), abs(res["pseudo_heritability"] - ref["pseudo_heritability"])]
    worst = [max(w, o) for w, o in zip(worst, out)]
    print("%5d %5d %9.2e | %10.2e %10.2e | %10.2e %10.2e" % (n, len(snps), ref["pseudo_heritability"], *out), flush=True)
print("worst %-13s | %10.2e %10.2e | %10.2e %10.2e" % ("", *worst))

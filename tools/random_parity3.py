#!/usr/bin/env python3
"""Third randomised sweep: the variance-component estimates alone -- REML and ML, through the eigendecomposition-free route
(band reduction + interpolated search) and the eigen route -- against the oracle, over kinships of different make (IBS of
many / few SNPs, GRM, with duplicated individuals) and phenotypes whose heritability runs from 0 to 0.99 (optimum at either
end of the grid), with 0-3 cofactors.  Checker only.   python tools/random_parity3.py [cases] [seed]"""
import os, sys, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mixmogam_amd import _lib, kinship, linear_models as lm
from oracle import emmax_oracle as orc
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = _lib.get_context()
rng = np.random.RandomState(seed)
worst, fails = {}, 0


def note(key, val, tol, what):
    global fails
    worst[key] = max(worst.get(key, 0.0), val)
    if not val <= tol:
        fails += 1
        print("  FAIL %-22s %.3e > %.1e   %s" % (key, val, tol, what), flush=True)


for c in range(cases):
    n = int(rng.choice([rng.randint(30, 255), rng.randint(256, 800), rng.randint(800, 2200)]))
    m = int(rng.choice([rng.randint(5, 60), rng.randint(200, 3000)]))
    snps = (rng.random_sample((m, n)) < rng.uniform(0.05, 0.95, m)[:, None]).astype(np.int8)
    if rng.rand() < 0.3:
        dst = rng.choice(n, n // 4, replace=False)
        snps[:, dst] = snps[:, rng.randint(0, n, n // 4)]
    snps = snps[snps.std(1) > 0]
    if len(snps) < 2:
        continue
    kk = rng.choice(["ibs", "grm"])
    K = orc.calc_ibs_kinship(snps) if kk == "ibs" else orc.calc_ibd_kinship(snps)
    h2 = float(rng.choice([0.0, 0.2, 0.6, 0.9, 0.99]))
    Ks = orc.scale_k(K)
    w, U = np.linalg.eigh(Ks)
    gpart = U @ (np.sqrt(np.maximum(w, 0)) * rng.standard_normal(n))
    y = np.sqrt(h2) * gpart / max(gpart.std(), 1e-12) + np.sqrt(1 - h2) * rng.standard_normal(n)
    q = int(rng.choice([0, 1, 3]))
    cof = [list(rng.standard_normal(n)) for _ in range(q)]
    X = np.hstack([np.ones((n, 1))] + [np.asarray(cv).reshape(n, 1) for cv in cof])
    what = "case %d: n=%d m=%d %s h2=%.2f q=%d" % (c, n, len(snps), kk, h2, q)
    try:
        for method in ("REML", "ML"):
            ref = orc.get_estimates(y, X, Ks, method=method)
            for route in ("free", "eigen"):
                if route == "free" and n <= lm.EIGEN_FREE_MIN_N:
                    continue
                lmm = lm.LinearMixedModel(list(y), ctx=ctx)
                lmm.add_random_effect(K)
                for cv in cof:
                    lmm.add_factor(cv)
                if route == "free":
                    got = lmm.get_estimates_eigen_free(method=method)
                    got.pop("reml").close()
                else:
                    got = lmm.get_estimates(lmm._get_eigen_L_(), method=method)
                tag = "%s %s" % (method, route)
                note(tag + " max_ll", abs(got["max_ll"] - ref["max_ll"]) / max(1.0, abs(ref["max_ll"])), 1e-9, what)
                note(tag + " h2", abs(got["pseudo_heritability"] - ref["pseudo_heritability"]), 1e-6, what + " ref delta %.3e" % ref["delta"])
                note(tag + " ve", abs(got["ve"] - ref["ve"]) / max(abs(ref["ve"]), 1e-12), 1e-5, what)
    except Exception as e:                                     # noqa: report and continue
        fails += 1
        print("  EXCEPTION %s: %s: %s" % (what, type(e).__name__, str(e)[:300]), flush=True)
        traceback.print_exc(limit=4)
print("worst over %d cases: %s" % (cases, {k: "%.2e" % v for k, v in sorted(worst.items())}))
print("failures: %d" % fails)
sys.exit(1 if fails else 0)

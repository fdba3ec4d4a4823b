#!/usr/bin/env python3
"""The REML phase of emmax() by itself -- workspace from the kinship in HBM, band reduction, likelihood search -- timed call by
call, with the library's own stage times (MMG_REML_VERBOSE=1 on stderr).  Run once per setting of MMG_REML_FINE_GRID.
    python tools/reml_phase_ab.py [N] [calls] [structured|hash]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mixmogam_amd import _lib, kinship, linear_models as lm
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = _lib.get_context()
kind = sys.argv[3] if len(sys.argv) > 3 else "structured"   # "hash": unrelated individuals, 100 causal SNPs (bench.py's data)
rng = np.random.RandomState(3)
if kind == "hash":
    g = ctx.geno(M=1000000, N=n).fill_hash(20240)
    causal = np.sort(rng.choice(1000000, 100, replace=False))
    gen = rng.exponential(1.0, size=100) @ np.asarray(g.download_rows(causal), dtype=np.float64)
    err = rng.normal(0, 1, size=n)
    y = gen + err * np.sqrt((0.2 / 0.8) * (np.var(gen, ddof=1) / np.var(err, ddof=1)))
    y = (y - y.mean()) / y.std()
else:
    g = ctx.geno(M=200000, N=n).fill_structured(20250, npop=3)
    y = rng.standard_normal(n) + np.asarray(g.download_rows(np.arange(5)), dtype=np.float64).sum(0)
K = kinship.calc_ibs_kinship(None, ctx=ctx, geno=g, keep_device=True)
for c in range(calls):
    mdl = lm.LinearMixedModel(list(y), ctx=ctx)
    mdl.add_random_effect(K)
    t0 = time.time()
    res = mdl.get_estimates_eigen_free()
    t1 = time.time()
    res.pop("reml").close()
    t2 = time.time()
    print("call %d: estimates %.2f ms (+ close %.2f ms), %d device call(s), %d variance ratios, delta %.6g" % (
        c, (t1 - t0) * 1e3, (t2 - t1) * 1e3, res["n_device_calls"], res["n_factorisations"], res["delta"]), flush=True)

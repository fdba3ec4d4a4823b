#!/usr/bin/env python3
"""The band reduction of K with and without look-ahead (MMG_BAND_LOOKAHEAD is read once per process: one process per setting):
python tools/band_lookahead_ab.py N   -> seconds of the reduction (best of 3 workspaces) and the REML sums' agreement with the
per-delta Cholesky route."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixmogam_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
ctx = _lib.get_context()
rng = np.random.RandomState(0)
from mixmogam_amd import kinship
g = ctx.geno(M=40 * N, N=N).fill_hash(20240)
K = kinship.calc_ibs_kinship(None, geno=g, ctx=ctx)         # an IBS kinship of 40 N random SNPs, as bench.py's
g.close()
y = rng.standard_normal(N)
X = np.column_stack([np.ones(N), rng.standard_normal(N)])
deltas = np.array([0.05, 1.0, 30.0])
best, first = 1e9, None
for rep in range(4):
    reml = ctx.reml(K, X, y)
    t0 = time.time()
    band = reml.sums(deltas, route="band")
    dt = time.time() - t0
    sec = reml.band_info()["seconds"]
    fb = reml.band_info()["householder_fallback"]
    if first is None:
        first = sec
    else:
        best = min(best, sec)
    if rep == 3:
        chol = reml.sums(deltas, route="chol")
        err = max(float(np.max(np.abs(band[i] - chol[i]) / np.maximum(np.abs(chol[i]), 1.0))) for i in range(4))
    reml.close()
print("N=%d MMG_BAND_LOOKAHEAD=%s: band reduction %.2f ms (first of the process %.2f), sums vs the Cholesky route %.2e, fallback %s"
      % (N, os.environ.get("MMG_BAND_LOOKAHEAD", "1"), best * 1e3, first * 1e3, err, fb))

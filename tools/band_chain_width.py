#!/usr/bin/env python3
"""How the cost of one device call of the band route (mmg_reml_sums: a banded factorisation, substitutions and trace recurrence
per variance ratio, one workgroup each) grows with the number of variance ratios in the call.
    python tools/band_chain_width.py [N]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mixmogam_amd import _lib, kinship
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
ctx = _lib.get_context()
g = ctx.geno(M=100000, N=n).fill_structured(20250, npop=3)
K = kinship.calc_ibs_kinship(None, ctx=ctx, geno=g)
rng = np.random.RandomState(1)
y = rng.standard_normal(n)
reml = ctx.reml(K, np.ones((n, 1)), y)
reml.sums(np.exp(np.linspace(-10, 10, 51)))            # band reduction + first call
for m in (1, 16, 51, 101, 151, 201, 221, 256, 301, 401):
    d = np.exp(np.linspace(-10.5, 10.5, m))
    ts = []
    for _ in range(4):
        t0 = time.time(); reml.sums(d); ts.append(time.time() - t0)
    print("N=%d  %3d variance ratios per call: %.2f ms (%s)" % (n, m, min(ts) * 1e3, " ".join("%.2f" % (t * 1e3) for t in ts)), flush=True)
reml.close()

#!/usr/bin/env python3
"""End-to-end latency of lm.emmax() on device-resident genotypes at the headline shape (N=5000, M=1e6), stage by
stage, with the scan model built on the host from H_sqrt_inv (round-1 route) and on the device from K and delta."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixmogam_amd import _lib, kinship, linear_models as lm
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
ctx = _lib.get_context()
g = ctx.geno(M=M, N=N).fill_hash(20240)
t0 = time.time()
K = kinship.calc_ibs_kinship(None, ctx=ctx, geno=g)
print("calc_ibs_kinship: %.3f s" % (time.time() - t0))
rng = np.random.RandomState(1)
y = rng.standard_normal(N) + g.download_rows([5])[0]
for flag in (False, True, True):
    lm.DEVICE_SCAN_MODEL = flag
    t0 = time.time()
    res = lm.emmax(g, list(y), K, ctx=ctx)
    dt = time.time() - t0
    print("emmax() device_model=%s: %.3f s  %s  min p %.3e" % (flag, dt, {k: round(v, 3) for k, v in res['timings'].items()},
                                                             res['ps'].min()))
    if flag is False:
        ref = res['ps'].copy()
print("max rel p diff between the two routes: %.2e" % float(np.max(np.abs(res['ps'] / ref - 1))))

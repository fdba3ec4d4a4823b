#!/usr/bin/env python3
"""Where the eigendecomposition-free route of emmax_f_test starts to pay: lm.emmax() on device-resident genotypes by both
routes (rocSOLVER dsyevd + REML from eig_L  |  band reduction of K + banded REML + device scan model) over a range of N.
    python tools/reml_vs_eigh.py [M] [N ...]        -> one line per N; linear_models.EIGEN_FREE_MIN_N is set from this"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixmogam_amd import _lib, kinship, linear_models as lm
M = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
Ns = [int(a) for a in sys.argv[2:]] or [300, 500, 1000, 1500, 2000, 3000, 4096, 5000, 8192]
ctx = _lib.get_context()
for N in Ns:
    g = ctx.geno(M=M, N=N).fill_structured(7, npop=3)
    K = kinship.calc_ibs_kinship(None, ctx=ctx, geno=g)
    rng = np.random.RandomState(1)
    y = rng.standard_normal(N) + g.download_rows([5])[0] + (K @ rng.standard_normal(N)) / np.sqrt(N)
    out = {}
    for name, min_n in (("eigen", 1 << 30), ("eigen_free", 0)):
        lm.EIGEN_FREE_MIN_N = min_n
        ts = []
        for rep in range(4):
            t0 = time.time()
            res = lm.emmax(g, list(y), K, ctx=ctx)
            ts.append(time.time() - t0)
        out[name] = (min(ts[1:]), res)
    a, b = out["eigen"][1], out["eigen_free"][1]
    print("N=%5d M=%d: eigen %.4f s (eig_L %.4f)  eigen_free %.4f s (reml %.4f)  ratio %.2f   delta %.5e / %.5e  max rel p diff %.1e"
          % (N, M, out["eigen"][0], a['timings']['eig_L'], out["eigen_free"][0], b['timings']['reml'], out["eigen"][0] / out["eigen_free"][0],
             1 / a['pseudo_heritability'] - 1, 1 / b['pseudo_heritability'] - 1, float(np.max(np.abs(a['ps'] / b['ps'] - 1)))), flush=True)
    g.close()

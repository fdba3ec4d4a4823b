import os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from mixmogam_amd import _lib, kinship, linear_models as lm
N = 5000
ctx = _lib.get_context()
g = ctx.geno(M=200000, N=N); g.fill_hash(20240, m_global0=0, thr16=32768)
K = kinship.scale_k(ctx.kinship_ibs_counts(g).astype(np.float64) / (2.0 * 200000) + 0.5)
y = np.random.RandomState(1).standard_normal(N)
for rep in range(3):
    lmm = lm.LinearMixedModel(y, ctx=ctx); lmm.add_random_effect(K)
    t0 = time.time(); est = lmm.get_estimates_eigen_free(); t1 = time.time()
    prep = lmm.scan_model_eigen_free(est); t2 = time.time()
    est.pop('reml').close()
    print("rep %d: REML %.1f ms, scan model %.1f ms" % (rep, 1e3*(t1-t0), 1e3*(t2-t1)), flush=True)

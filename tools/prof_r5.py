#!/usr/bin/env python3
"""Driver for the rocprofv3 passes over the kernels written in rounds 4-5: the band reduction of K and the
banded per-delta kernels (two eigendecomposition-free emmax() calls on a resident store), the one-pass exact GRM, the fp64
exact tier of the scan (a kinship of 12 genotype classes), the permutation plan from the Cholesky root.  Nothing else runs
(no rocSOLVER dispatch storms).     python3 tools/prof_r5.py [N] [M]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mixmogam_amd import _lib, kinship, linear_models as lm

N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 400000
ctx = _lib.get_context()
g = ctx.geno(M=M, N=N).fill_hash(20240)
rng = np.random.RandomState(1)
y = rng.standard_normal(N)
K = kinship.calc_ibs_kinship(None, geno=g, ctx=ctx, keep_device=True)
for _ in range(2):
    lmm = lm.LinearMixedModel(list(y), ctx=ctx)
    lmm.add_random_effect(K)
    r = lmm.emmax_f_test(g, emma_num=0)
print("emmax phases", r["timings"], "h2", r["pseudo_heritability"])
acc = ctx.kinship_accumulator(N)
for _ in range(2):
    acc.add_grm(g)
print("grm ms", ctx.kernel_ms("grm"))
acc.close()
# permutation plan + after-scan test on the Cholesky root
idx = np.array([np.random.RandomState(7 + p).permutation(N) for p in range(256)])
est = lmm.get_estimates_eigen_free()
prep = lmm.scan_model_eigen_free(est)
lp = lm.LinearMixedModel(list(y), ctx=ctx)
lp.random_effects = lmm.random_effects
pp = lp.perm_prepare(None, num_perm=len(idx), perm_idx=idx, reml=est["reml"], delta=est["delta"])
plan = est["reml"].perm_plan(est["delta"], pp["Ys"], pp["h0_rss"])
est["reml"].close()
print("perm min rss", float(plan.run(g).min()))
plan.close()
K.close()
# the exact tier: every individual is one of 12 genotype vectors
n2, m2 = 1500, 3000
proto = (rng.random_sample((m2, 12)) < 0.4).astype(np.int8)
s2 = proto[:, rng.randint(0, 12, n2)]
s2 = s2[s2.std(1) > 0]
y2 = rng.standard_normal(n2) + s2[3]
r2 = lm.emmax(s2, list(y2), kinship.calc_ibs_kinship(s2, ctx=ctx), ctx=ctx)
print("exact tier:", ctx.scan_last_stats())

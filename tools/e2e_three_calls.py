#!/usr/bin/env python3
"""Three lm.emmax()-equivalent calls on resident genotypes (kinship in HBM) -- the workload of the kernel trace that shows which
kernels the default route launches (no Cijk_* / rocblas_* / rocsolver::* rows).   python3 tools/e2e_three_calls.py [N] [M]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mixmogam_amd import _lib, kinship, linear_models as lm
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
ctx = _lib.get_context()
g = ctx.geno(M=M, N=N).fill_hash(20240)
y = np.random.RandomState(1).standard_normal(N)
host_k = os.environ.get("MMG_E2E_HOST_K") == "1"
K = kinship.calc_ibs_kinship(None, geno=g, ctx=ctx, keep_device=not host_k)
for i in range(3):
    t0 = time.time()
    lmm = lm.LinearMixedModel(list(y), ctx=ctx)
    lmm.add_random_effect(K)
    r = lmm.emmax_f_test(g, emma_num=0)
    print("call %d: %.4f s  %s  min p %.3e" % (i, time.time() - t0, {k: round(v, 4) for k, v in r["timings"].items()}, r["ps"].min()), flush=True)

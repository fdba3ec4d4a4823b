#!/usr/bin/env python3
"""Binary against 0/1/2 genotype stores on the three paths that have a fused form for binary stores only ("measure
and state today's ratio first"): exact GRM (one-pass four-plane kernel vs one GEMM per plane, five planes), IBS kinship (FP4 twin vs
two indicator products with their image passes for 'diploid_int'), EMMAX scan (linear terms in the GEMM vs the finalize sweep).
    python tools/diploid_ratio.py [N] [M]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixmogam_amd import _lib, kinship, linear_models as lm
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
ctx = _lib.get_context()
rng = np.random.RandomState(0)


def timed(f, reps=3):
    best = 1e9
    for _ in range(reps):
        t0 = time.time(); f(); best = min(best, time.time() - t0)
    return best * 1e3


stores = {}
stores["binary"] = ctx.geno(M=M, N=N).fill_hash(20240)
# 0/1/2 with Hardy-Weinberg-like frequencies from two binary draws: packed 2-bit rows, expanded on the device
codes = (rng.random_sample((M, N)) < 0.5).astype(np.uint8) + (rng.random_sample((M, N)) < 0.5).astype(np.uint8)
g2 = ctx.geno(M=M, N=N)
g2.upload_packed(_lib.pack_genotypes(codes.astype(np.int8), 2), 2)
del codes
stores["diploid 0/1/2"] = g2
y = rng.standard_normal(N)
for name, g in stores.items():
    acc = ctx.kinship_accumulator(N)
    acc.add_grm(g)
    t_grm = timed(lambda: acc.add_grm(g))
    grm_ms = ctx.kernel_ms("grm")
    acc.close()
    if name == "binary":
        t_ibs = timed(lambda: kinship.calc_ibs_kinship(None, ctx=ctx, geno=g, scaled=False))
    else:
        # the device route of calc_ibs_kinship('diploid_int') (mmg_kinship_ibs_diploid_f64; round 6: one stacked FP4 GEMM) against
        # the binary store's counts call; MMG_IBS_DIPLOID_FUSED=0: the two indicator products of round 5
        t_ibs = timed(lambda: kinship.calc_ibs_kinship(None, snps_data_format='diploid_int', ctx=ctx, geno=g, scaled=False))
    K = kinship.calc_ibs_kinship(None, ctx=ctx, geno=g) if name == "binary" else kinship.calc_ibs_kinship(None, snps_data_format='diploid_int', ctx=ctx, geno=g)
    lmm = lm.LinearMixedModel(list(y), ctx=ctx)
    lmm.add_random_effect(K)
    est = lmm.get_estimates_eigen_free()
    prep = lmm.scan_model_eigen_free(est)
    est.pop("reml").close()
    ctx.scan(g, prep["h0_rss"], prep["n_p"])
    t_scan = timed(lambda: ctx.scan(g, prep["h0_rss"], prep["n_p"]))
    print("%-14s N=%d M=%d: exact GRM %.1f ms wall (GEMMs %.1f)   IBS counts %.1f ms   scan %.1f ms (GEMM %.2f + finalize %.2f)"
          % (name, N, M, t_grm, grm_ms, t_ibs, t_scan, ctx.kernel_ms("scan_quad"), ctx.kernel_ms("scan_finalize")), flush=True)

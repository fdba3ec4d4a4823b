#!/usr/bin/env python3
"""What the number of weight digit planes of the exact GRM (mmg_kin_acc_add_grm) does downstream: K and the EMMAX p-values
with MMG_GRM_PLANES = 3 / 4 against 5 planes, on Bernoulli(0.5) genotypes (all weights equal) and on structured ones (allele
frequencies 0.02 .. 0.98: weights spread over a factor ~12).   python tools/grm_planes_check.py [N] [M]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixmogam_amd import _lib, linear_models as lm
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
ctx = _lib.get_context()
for name in ("bernoulli", "structured"):
    g = ctx.geno(M=M, N=N)
    g.fill_hash(20240) if name == "bernoulli" else g.fill_structured(20250, npop=3)
    mean, sd = g.snp_stats()
    w = 1.0 / sd ** 2
    rng = np.random.RandomState(1)
    out = {}
    for planes in ("5", "4", "3"):
        os.environ["MMG_GRM_PLANES"] = planes
        acc = ctx.kinship_accumulator(N)
        t0 = time.time()
        acc.add_grm(g)
        dt = time.time() - t0
        acc.scale_k()
        K, cnt = acc.fetch()
        acc.close()
        if planes == "5":
            y = rng.standard_normal(N) + g.download_rows([5])[0] + 1.5 * (K @ rng.standard_normal(N)) / np.sqrt(N)
        res = lm.emmax(g, list(y), K, ctx=ctx)
        out[planes] = (K, res, dt)
    del os.environ["MMG_GRM_PLANES"]
    K5, r5, _ = out["5"]
    print("%s: N=%d M=%d  weights max/min %.2f  delta %.6e  min p %.2e" % (name, N, M, w.max() / w.min(), 1 / r5['pseudo_heritability'] - 1, r5['ps'].min()))
    for planes in ("4", "3"):
        K, r, dt = out[planes]
        print("   %s planes: add_grm %.3f s  max |K - K5| / max |K5| %.2e   delta rel diff %.2e   max rel p diff %.2e (p > 1e-300)"
              % (planes, dt, np.max(np.abs(K - K5)) / np.max(np.abs(K5)), abs((1 / r['pseudo_heritability'] - 1) / (1 / r5['pseudo_heritability'] - 1) - 1),
                 float(np.max(np.abs(r['ps'][r5['ps'] > 1e-300] / r5['ps'][r5['ps'] > 1e-300] - 1)))), flush=True)
    g.close()

#!/usr/bin/env python3
"""lm.emmax() end to end on device-resident genotypes at N (default: the headline shape), by route: the eigen route
(rocSOLVER dsyevd) against the eigendecomposition-free route (band reduction of K + device scan model), stage by stage,
with a host-side profile of the glue.   python tools/e2e_routes.py [N] [M] [--profile]   (MMG_REML_VERBOSE=1: stages)"""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixmogam_amd import _lib, kinship, linear_models as lm
args = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(args[0]) if len(args) > 0 else 5000
M = int(args[1]) if len(args) > 1 else 1000000
ctx = _lib.get_context()
g = ctx.geno(M=M, N=N).fill_hash(20240)
K = kinship.calc_ibs_kinship(None, ctx=ctx, geno=g)
rng = np.random.RandomState(1)
y = rng.standard_normal(N) + g.download_rows([5])[0]
ref = None
routes = (("eigen_free", 0),) if "--free-only" in sys.argv else (("eigen", 1 << 30), ("eigen_free", 0))
for name, min_n in routes:
    lm.EIGEN_FREE_MIN_N = min_n
    for rep in range(3):
        t0 = time.time()
        res = lm.emmax(g, list(y), K, ctx=ctx)
        dt = time.time() - t0
        print("emmax() route=%s: %.3f s  %s  delta %.6e min p %.3e" % (name, dt, {k: round(v, 4) for k, v in res['timings'].items()},
                                                                      1.0 / res['pseudo_heritability'] - 1.0, res['ps'].min()), flush=True)
    if "--profile" in sys.argv:
        pr = cProfile.Profile()
        pr.enable()
        lm.emmax(g, list(y), K, ctx=ctx)
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
    if ref is None:
        ref = res['ps'].copy()
print("max rel p diff between the two routes: %.2e" % float(np.max(np.abs(res['ps'] / ref - 1))))

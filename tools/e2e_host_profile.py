#!/usr/bin/env python3
"""Where the wall time of one lm.emmax()-equivalent call on resident genotypes goes on the HOST side (cProfile of a warm call
at BASELINE config 3: N = 5000, M = 10^6), next to the device events of the same call.
    python tools/e2e_host_profile.py [N] [M]"""
import cProfile, io, os, pstats, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mixmogam_amd import _lib, kinship, linear_models as lm
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
ctx = _lib.get_context()
g = ctx.geno(M=m, N=n).fill_structured(20250, npop=3)
rng = np.random.RandomState(3)
K = kinship.calc_ibs_kinship(None, ctx=ctx, geno=g)
y = rng.standard_normal(n) + np.asarray(g.download_rows(np.arange(5)), dtype=np.float64).sum(0)


def call():
    mdl = lm.LinearMixedModel(list(y), ctx=ctx)
    mdl.add_random_effect(K)
    return mdl.emmax_f_test(g, emma_num=0)


for _ in range(2):
    t0 = time.time(); r = call(); dt = time.time() - t0
print("warm call %.1f ms  timings %s" % (dt * 1e3, {k: round(v * 1e3, 1) for k, v in r["timings"].items()}))
pr = cProfile.Profile()
pr.enable(); r = call(); pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue()[:6000])
for name in ("scan", "scan_model", "reml", "grm", "ibs"):
    try:
        print("kernel_ms(%s) = %.2f" % (name, ctx.kernel_ms(name)))
    except Exception as e:
        print("kernel_ms(%s): %s" % (name, e))

#!/usr/bin/env python3
"""Every public entry point of the reference's call surface at the shapes of BASELINE configs[0] / [1], from HOST arrays, second
call of each (first calls load code objects): a table to spot a path that costs seconds where milliseconds are due.
    python tools/surface_timing.py [--profile NAME]"""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixmogam_amd import _lib, kinship, linear_models as lm
ctx = _lib.get_context()
prof = sys.argv[sys.argv.index("--profile") + 1].split(",") if "--profile" in sys.argv else []
shapes = [(199, 214000), (1000, 500000)]
if "--big" in sys.argv:
    shapes = [(5000, 200000)]
for n, m in shapes:
    rng = np.random.RandomState(n)
    freq = rng.uniform(0.1, 0.9, m)
    snps = (rng.random_sample((m, n)) < freq[:, None]).astype(np.int8)
    y = rng.standard_normal(n) + snps[7] + 0.5 * snps[11]
    ys = np.vstack([y, rng.standard_normal(n) + snps[100], rng.standard_normal(n) + snps[200], rng.standard_normal(n)])
    cof = [list(rng.standard_normal(n)), list(rng.standard_normal(n))]   # the reference's callers: a list of vectors
    K = kinship.calc_ibs_kinship(snps, ctx=ctx)
    snp_list = [s for s in snps[:20000]]                      # the reference's callers pass lists of arrays
    cases = [
        ("kinship.calc_ibs_kinship", lambda: kinship.calc_ibs_kinship(snps, ctx=ctx)),
        ("kinship.calc_ibd_kinship", lambda: kinship.calc_ibd_kinship(snps, ctx=ctx)),
        ("kinship.calc_ibs_kinship(diploid_int)", lambda: kinship.calc_ibs_kinship(snps, snps_data_format='diploid_int', ctx=ctx)),
        ("lm.emmax", lambda: lm.emmax(snps, list(y), K, ctx=ctx)),
        ("lm.emmax(list of 20,000 arrays)", lambda: lm.emmax(snp_list, list(y), K, ctx=ctx)),
        ("lm.emmax(cofactors)", lambda: lm.emmax(snps, list(y), K, cofactors=cof, ctx=ctx)),
        ("lm.emmax(with_betas)", lambda: lm.emmax(snps, list(y), K, with_betas=True, ctx=ctx)),
        ("lm.emmax(emma_num=100)", lambda: lm.emmax(snps, list(y), K, emma_num=100, ctx=ctx)),
        ("lm.emma (first 2000 SNPs)", lambda: lm.emma(snps[:2000], list(y), K, ctx=ctx)),
        ("lm.linear_model", lambda: lm.linear_model(snps, list(y), ctx=ctx)),
        ("lm.emmax_multi (4 phenotypes)", lambda: lm.emmax_multi(snps, ys, K, ctx=ctx)),
        ("lm.emmax_perm_test (100 perms)", lambda: lm.emmax_perm_test(snps, list(y), K, num_perm=100, ctx=ctx)),
        ("lm.mlmm (3 steps)", lambda: lm.mlmm(list(y), K, snps=snps, positions=np.arange(m), chromosomes=np.ones(m, dtype=int),
                                             num_steps=3, ctx=ctx)),
        ("lm.get_emma_reml_estimates", lambda: lm.get_emma_reml_estimates(list(y), K, ctx=ctx)),
    ]
    print("---- N = %d, M = %d (host arrays)" % (n, m), flush=True)
    for name, fn in cases:
        try:
            fn()
            t0 = time.time(); fn(); dt = time.time() - t0
            print("  %-44s %9.1f ms" % (name, 1e3 * dt), flush=True)
            if any(p_ in name for p_ in prof):
                pr = cProfile.Profile(); pr.enable(); fn(); pr.disable()
                pstats.Stats(pr).sort_stats("tottime").print_stats(14)
        except Exception as e:                                 # noqa: a table of timings: report and go on
            print("  %-44s FAILED: %s: %s" % (name, type(e).__name__, str(e)[:150]), flush=True)

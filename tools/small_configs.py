#!/usr/bin/env python3
"""BASELINE configs[0] and [1] as a user of the reference runs them (examples.py): genotypes on the HOST, kinship.calc_ibs_kinship
then linear_models.emmax -- wall time per call, stage times, and where the host spends it.
    python tools/small_configs.py [--profile]"""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixmogam_amd import _lib, kinship, linear_models as lm
ctx = _lib.get_context()
for name, n, m in (("configs[0] shape (A. thaliana: 199 x 214,000)", 199, 214000), ("configs[1] (1000 x 500,000)", 1000, 500000),
                   ("2000 x 500,000", 2000, 500000)):
    rng = np.random.RandomState(n)
    freq = rng.uniform(0.1, 0.9, m)
    snps = (rng.random_sample((m, n)) < freq[:, None]).astype(np.int8)
    y = rng.standard_normal(n) + snps[7] + 0.5 * snps[11]
    for rep in range(3):
        t0 = time.time()
        K = kinship.calc_ibs_kinship(snps, ctx=ctx)
        t1 = time.time()
        res = lm.emmax(snps, list(y), K, ctx=ctx)
        t2 = time.time()
        print("%s: kinship %.1f ms, emmax() %.1f ms %s  min p %.2e" % (name, 1e3 * (t1 - t0), 1e3 * (t2 - t1),
              {k: round(1e3 * v, 1) for k, v in res['timings'].items()}, res['ps'].min()), flush=True)
    if "--profile" in sys.argv:
        pr = cProfile.Profile(); pr.enable()
        K = kinship.calc_ibs_kinship(snps, ctx=ctx)
        lm.emmax(snps, list(y), K, ctx=ctx)
        pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(14)

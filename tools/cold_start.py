#!/usr/bin/env python3
"""First call of a fresh process, by route: what a user who runs one analysis per process waits for beyond the arithmetic.
    python tools/cold_start.py            (runs each case in a process of its own, in the order given)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) == 1:
    for case in ("1000", "199", "199", "1000"):
        subprocess.run([sys.executable, os.path.abspath(__file__), case], check=False)
    sys.exit(0)
t_start = time.time()
import numpy as np
sys.path.insert(0, ROOT)
from mixmogam_amd import _lib, kinship, linear_models as lm
n = int(sys.argv[1]); m = 100000
ctx = _lib.get_context()
t_ctx = time.time()
rng = np.random.RandomState(n)
snps = (rng.random_sample((m, n)) < rng.uniform(0.1, 0.9, m)[:, None]).astype(np.int8)
y = rng.standard_normal(n) + snps[7]
t0 = time.time()
K = kinship.calc_ibs_kinship(snps, ctx=ctx)
t1 = time.time()
res = lm.emmax(snps, list(y), K, ctx=ctx)
t2 = time.time()
res2 = lm.emmax(snps, list(y), K, ctx=ctx)
t3 = time.time()
print("N=%d: import + context %.2f s; first kinship %.3f s; first emmax() %.3f s %s; second emmax() %.3f s"
      % (n, t_ctx - t_start, t1 - t0, t2 - t1, {k: round(v, 3) for k, v in res['timings'].items()}, t3 - t2), flush=True)

#!/usr/bin/env python3
"""Stress of the adaptive scan path (GPU box): random shapes, alphabets and matrices; default model vs explicit
4 planes -- max relative p difference, statistics, no crash."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mixmogam_amd import _lib
ctx = _lib.Context(0)
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = 0.0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    n = int(rng.choice([17, 60, 255, 256, 257, 700, 1300, 2600]))
    m = int(rng.choice([1, 5, 255, 256, 257, 1000, 4097, 20000]))
    hi = int(rng.choice([2, 3]))
    snps = rng.randint(0, hi, size=(m, n)).astype(np.int8)
    B = rng.standard_normal((n, 12)) / 3
    A = np.eye(n) * rng.uniform(0.5, 3) + (B @ B.T) / n * rng.uniform(0.1, 10)
    A += 0.5 * (lambda R: R + R.T)(rng.standard_normal((n, n)) * rng.uniform(0, 0.05))
    w = rng.standard_normal(n)
    g = ctx.geno(snps)
    h0 = float(rng.uniform(1, 1e6))
    ctx.scan_set_model(A, w, 4); full = ctx.scan(g, h0, max(2, n - 2))
    ctx.scan_set_model(A, w, 0); ada = ctx.scan(g, h0, max(2, n - 2)); st = ctx.scan_last_stats()
    ok = (full["ps"] > 1e-290) & np.isfinite(full["ps"])
    d = float(np.max(np.abs(ada["ps"][ok] / full["ps"][ok] - 1))) if ok.any() else 0.0
    worst = max(worst, d)
    print("n=%4d m=%5d alphabet 0..%d  max rel p diff %.2e  %s" % (n, m, hi - 1, d, st), flush=True)
    assert d < 1e-6
    g.close()
print("worst", worst)

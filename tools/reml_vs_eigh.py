"""REML likelihood sums at size N: band route (one reduction + 57 deltas) against the eigendecomposition it replaces.
   python tools/reml_vs_eigh.py N [N ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixmogam_amd import _lib
ctx = _lib.get_context()
for N in [int(a) for a in sys.argv[1:]]:
    rng = np.random.RandomState(0)
    B = rng.standard_normal((N, 64))
    K = B @ B.T / 64 + 0.5 * np.eye(N)
    y = rng.standard_normal(N)
    X = np.ones((N, 1))
    deltas = np.exp(np.linspace(-10, 10, 51))
    for rep in range(2):
        reml = ctx.reml(K, X, y)
        t0 = time.time()
        reml.sums(deltas, route="band")
        for k in range(6):
            reml.sums(deltas[20 + k:21 + k], route="band")
        t_band = time.time() - t0
        t0 = time.time()
        reml.scan_model(1.0)
        t_model = time.time() - t0
        reml.close()
        t0 = time.time()
        ctx.eigh(K)
        t_eigh = time.time() - t0
        print("N=%d rep %d: band route (reduction + 51 + 6 deltas) %.3f s, scan model %.3f s | eigh %.3f s (kernel %.1f ms)"
              % (N, rep, t_band, t_model, t_eigh, ctx.kernel_ms("eigh")), flush=True)

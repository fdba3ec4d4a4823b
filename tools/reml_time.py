"""REML likelihood sums at size N by both routes of mmg_reml_sums_ex: one Cholesky factorisation per delta against one
band reduction for all deltas (csrc/reml_band.hip).   python tools/reml_time.py N [ndeltas]   (MMG_REML_VERBOSE=1: stages)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixmogam_amd import _lib
N = int(sys.argv[1])
nd = int(sys.argv[2]) if len(sys.argv) > 2 else 51
ctx = _lib.get_context()
rng = np.random.RandomState(0)
B = rng.standard_normal((N, 64))
K = B @ B.T / 64 + 0.5 * np.eye(N)
y = rng.standard_normal(N)
X = np.ones((N, 1))
t0 = time.time()
reml = ctx.reml(K, X, y)
print("N=%d create %.2f s" % (N, time.time() - t0), flush=True)
deltas = np.exp(np.linspace(-10, 10, nd))
t0 = time.time()
band = reml.sums(deltas, route="band")
print("band route, %d deltas, first call (reduction + band kernels): %.3f s" % (nd, time.time() - t0), flush=True)
t0 = time.time()
band2 = reml.sums(deltas, route="band")
print("band route, %d deltas, later call: %.3f s" % (nd, time.time() - t0), flush=True)
t0 = time.time()
one = reml.sums(deltas[nd // 2:nd // 2 + 1], route="band")
print("band route, 1 delta: %.3f s" % (time.time() - t0), flush=True)
pick = [0, nd // 2, nd - 1]
t0 = time.time()
chol = reml.sums(deltas[pick], route="chol")
dt = (time.time() - t0) / len(pick)
print("cholesky route: %.3f s per delta -> %.1f s for %d" % (dt, dt * nd, nd), flush=True)
for i in range(4):
    err = np.max(np.abs(band[i][pick] - chol[i]) / np.maximum(np.abs(chol[i]), 1.0))
    print("  s%d: max rel diff band vs cholesky %.2e" % (i + 1, err))

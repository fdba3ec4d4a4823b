import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from mixmogam_amd import _lib, linear_models as lm
N = int(sys.argv[1])
ctx = _lib.get_context()
rng = np.random.RandomState(0)
B = rng.standard_normal((N, 64))
K = B @ B.T / 64 + 0.5 * np.eye(N)
y = rng.standard_normal(N)
X = np.ones((N, 1))
t0 = time.time()
reml = ctx.reml(K, X, y)
print("create %.2f s" % (time.time() - t0))
for d in (0.5, 2.0):
    t0 = time.time()
    out = reml.sums([d])
    print("delta %.1f: %.3f s" % (d, time.time() - t0), [float(o[0]) for o in out[:4]])

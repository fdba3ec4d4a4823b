"""A/B of the XCD group shapes of the rotation / permutation GEMMs (MMG_ROT_GV, MMG_PERM_GV are read once per process:
run one process per setting).  usage: gv_sweep.py N M P"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixmogam_amd import _lib
N, M, P = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ctx = _lib.get_context()
g = ctx.geno(M=M, N=N).fill_hash(1)
rng = np.random.RandomState(0)
Q, _ = np.linalg.qr(rng.standard_normal((N, N)))
rot = ctx.rot(np.ascontiguousarray(Q.T), M)
ms = []
for _ in range(3):
    rot.load(g)
    ms.append(ctx.kernel_ms("rotate"))
rot.close()
H = Q * (1.0 + rng.rand(N))[:, None]
Ys = rng.standard_normal((N, P))
plan = ctx.perm_plan(H, Ys, float(N))
pm = []
for _ in range(3):
    plan.run(g)
    pm.append(ctx.kernel_ms("perm"))
print("ROT_GV=%s PERM_GV=%s: rotation %s ms, perm GEMM %s ms" % (os.environ.get("MMG_ROT_GV", "-"), os.environ.get("MMG_PERM_GV", "-"),
      ["%.2f" % x for x in ms], ["%.2f" % x for x in pm]), flush=True)

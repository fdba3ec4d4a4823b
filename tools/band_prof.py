"""Five band reductions of a structured kinship at size N (default 5000), for `rocprofv3 --kernel-trace --stats -- python3 tools/band_prof.py`:
per-kernel averages of the panel loop (csrc/reml_band.hip + dense64.hip).  Prints the reduction's own wall time per run."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixmogam_amd import _lib, kinship
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
ctx = _lib.get_context()
g = ctx.geno(M=200000, N=N)
g.fill_hash(20240, m_global0=0, thr16=32768)
K = kinship.scale_k(ctx.kinship_ibs_counts(g).astype(np.float64) / (2.0 * 200000) + 0.5)
y = np.random.RandomState(1).standard_normal(N)
REPS = int(os.environ.get("BAND_PROF_REPS", 6))
ms = []
for rep in range(REPS):
    rw = ctx.reml(K, np.ones((N, 1)), y)
    rw.sums(np.array([1.0]), route="band")
    ms.append(1e3 * rw.band_info()["seconds"])
    if REPS <= 6:
        print("run %d: band reduction %.3f ms" % (rep, ms[-1]), flush=True)
    rw.close()
print("N=%d, %d runs: min %.3f ms  median %.3f ms  (%s)" % (N, REPS, min(ms[1:]), sorted(ms[1:])[len(ms[1:]) // 2],
      " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith(("MMG_BAND", "MMG_HEAD")))), flush=True)

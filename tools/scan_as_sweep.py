"""The scan GEMM's workgroup grouping (MMG_SCAN_AS: SNP blocks per XCD cohort that share a digit-tile stream) at a size
where the model does not fit the Infinity Cache:  python tools/scan_as_sweep.py N M [AS ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixmogam_amd import _lib
N, M = int(sys.argv[1]), int(sys.argv[2])
sweep = [int(a) for a in sys.argv[3:]] or [4, 2, 8, 16]
ctx = _lib.get_context()
rng = np.random.RandomState(0)
B = rng.standard_normal((N, 64))
K = B @ B.T / 64 + 0.5 * np.eye(N)
y = rng.standard_normal(N)
reml = ctx.reml(K, np.ones((N, 1)), y)
del K
g = ctx.geno(M=M, N=N).fill_hash(20240)
for AS in sweep:
    os.environ["MMG_SCAN_AS"] = str(AS)
    t0 = time.time()
    h0, beta = reml.scan_model(1.0)
    tm = time.time() - t0
    ctx.scan(g, h0, N - 2, fetch=False)
    ms = []
    for _ in range(3):
        ctx.scan(g, h0, N - 2, fetch=False)
        ms.append(ctx.kernel_ms("scan_quad"))
    print("N=%d M=%d AS=%2d: scan model %.2f s, quad GEMM %s ms, %s" % (N, M, AS, tm, ["%.1f" % v for v in ms], ctx.scan_last_stats()), flush=True)

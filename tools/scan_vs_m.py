"""Wall time and kernel time of one EMMAX scan as a function of the store size (tools/scan_vs_m.py): what a chunked
pipeline pays per chunk beyond the GEMM itself."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixmogam_amd import _lib
N = 5000
ctx = _lib.get_context()
rng = np.random.RandomState(0)
A = rng.standard_normal((N, N)); A = A @ A.T / N
ctx.scan_set_model(A, rng.standard_normal(N), 0)
for M in (1000000, 400000, 200000, 100000, 50000, 25000):
    g = ctx.geno(M=M, N=N).fill_hash(20240)
    ctx.scan(g, 1000.0, N - 2, fetch=False)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); ctx.scan(g, 1000.0, N - 2, fetch=False); ts.append(time.perf_counter() - t0)
    print("M=%8d: wall %.3f ms (%.2f M SNPs/s), quad GEMM %.3f ms, finalize %.3f ms, %s" % (
        M, 1e3 * np.median(ts), M / np.median(ts) / 1e6, ctx.kernel_ms("scan_quad"), ctx.kernel_ms("scan_finalize"), ctx.scan_last_stats()), flush=True)
    g.close()

#!/usr/bin/env python3
"""Large-N validation of the scan path (C5-shaped sizes): synthetic symmetric A (no eigh), hash
genotypes; checks den of a handful of SNPs against a float64 host evaluation and reports SNPs/s.
usage: large_n_check.py N M [digits]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mixmogam_amd import _lib
N = int(sys.argv[1]); M = int(sys.argv[2]); D = int(sys.argv[3]) if len(sys.argv) > 3 else 4
ctx = _lib.Context(0)
print("host mem:", os.popen("free -g | sed -n 2p").read().strip())
rng = np.random.RandomState(0)
B = rng.standard_normal((N, 32)).astype(np.float64) / 6.0
t0 = time.time()
A = B @ B.T
A /= N
A[np.diag_indices(N)] += 1.0
w = rng.standard_normal(N)
print("built A %.1f s" % (time.time() - t0))
g = ctx.geno(M=M, N=N).fill_hash(20240)
t0 = time.time(); ctx.scan_set_model(A, w, D); print("set_model %.1f s" % (time.time() - t0))
for _ in range(2):
    out = ctx.scan(g, 1.0e9, N - 2, stats=True)
qms = ctx.kernel_ms("scan_quad"); fms = ctx.kernel_ms("scan_finalize")
idx = np.linspace(0, M - 1, 24).astype(int)
err = 0.0
for i in idx:
    s = g.download(int(i), 1)[0].astype(np.float64)
    err = max(err, abs(out["den"][i] / (s @ (A @ s)) - 1), abs(out["dot"][i] / (s @ w) - 1))
nJ = -(-N // 256)
exec_ops = 2.0 * D * 256.0 ** 3 * (nJ * (nJ + 1) / 2) * (-(-M // 256))
print({"N": N, "M": M, "digits": D, "scan_quad_ms": qms, "finalize_ms": fms, "snps_per_s_kernels": M / ((qms + fms) * 1e-3),
       "executed_int8_tops": exec_ops / (qms * 1e-3) / 1e12, "algorithmic_tflops": 2.0 * N * N * M / (qms * 1e-3) / 1e12,
       "max_rel_err_den_dot": err})

#!/usr/bin/env python
"""Timing of the eigen-rotated store and the multi-phenotype passes on the GPU box:
    python tools/multi_check.py [N] [M] [P]
prints the rotation GEMM time (int8 MFMA, 8 N^2 ops per SNP), the pass time per 8 phenotypes (HBM: 8 N bytes per
SNP) and checks a sample of p-values against a float64 host evaluation."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixmogam_amd import _lib, kinship, linear_models as lm  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
P = int(sys.argv[3]) if len(sys.argv) > 3 else 16
ctx = _lib.get_context()
g = ctx.geno(M=M, N=N)
g.fill_hash(20240, 0, 32768)
counts = ctx.kinship_ibs_counts(g)
K = kinship.scale_k(counts / (2.0 * M) + 0.5)
rng = np.random.RandomState(1)
ys = rng.standard_normal((P, N))
t0 = time.time()
lmm = lm.LinearMixedModel(ys[0], ctx=ctx)
lmm.add_random_effect(K)
eig_L = lmm._get_eigen_L_()
t_eig = time.time() - t0
t0 = time.time()
models, d, omega, G = lm._multi_models(ys, lmm.X, eig_L)
t_models = time.time() - t0
t0 = time.time()
rot = ctx.rot(eig_L['vectors'], M)
t_create = time.time() - t0
t0 = time.time()
rot.load(g)
t_load = time.time() - t0
rot_ms = ctx.kernel_ms("rotate")
h0 = np.array([m['h0_rss'] for m in models])
out = ctx.scan_multi(rot, d, omega, G, h0, N - 2)
t0 = time.time()
out = ctx.scan_multi(rot, d, omega, G, h0, N - 2)
t_multi = time.time() - t0
multi_ms = ctx.kernel_ms("scan_multi")
Npad = -(-N // 256) * 256
print("N=%d M=%d P=%d  eigh %.2fs  REML+null models %.3fs (%.1f ms/phenotype)  rot create %.2fs" %
      (N, M, P, t_eig, t_models, 1e3 * t_models / P, t_create))
print("rotation GEMM %.2f ms (%.0f int8 TOP/s executed; wall %.3fs)" %
      (rot_ms, 2.0 * 4 * Npad * (-(-N // 64) * 64) * M / (rot_ms * 1e-3) / 1e12, t_load))
nb = -(-P // 8)
print("multi passes: %.2f ms kernels for %d phenotypes (%d passes: %.2f ms/pass = %.2f TB/s of T; "
      "%.3f ms per phenotype-scan = %.1f M SNP-scans/s; wall incl. D2H of 3 x P x M doubles %.3fs)" %
      (multi_ms, P, nb, multi_ms / nb, 8.0 * N * M / (multi_ms / nb * 1e-3) / 1e12, multi_ms / P,
       M * P / (multi_ms * 1e-3) / 1e6, t_multi))
# sample check against float64 on the host
idx = rng.choice(M, 64, replace=False)
S = g.download_rows(idx).astype(np.float64)
V = np.asarray(eig_L['vectors'])
T = V @ S.T
err = 0.0
for p in range(P):
    aq = d[p] @ (T * T)
    den = aq - sum((G[p, c] @ T) ** 2 for c in range(G.shape[1]))
    dot = omega[p] @ T
    rss = h0[p] - dot * dot / den
    F = (h0[p] / rss - 1) * (N - 2)
    from scipy import stats
    ps = stats.f.sf(F, 1, N - 2)
    err = max(err, float(np.max(np.abs(out['ps'][p][idx] / ps - 1))))
print("max rel p error on a 64-SNP sample over all phenotypes vs float64 host: %.2e" % err)
# single-phenotype production scan of phenotype 0 for comparison
one = lm.emmax(g, list(ys[0]), K, ctx=ctx)
print("vs the quadratic-form GEMM scan of phenotype 0: max rel p diff %.2e" %
      float(np.max(np.abs(out['ps'][0] / one['ps'] - 1))))

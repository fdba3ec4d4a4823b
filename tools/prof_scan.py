#!/usr/bin/env python3
"""Minimal driver for rocprofv3 PMC passes: only this repo's kernels run (no rocSOLVER/rocBLAS
dispatch storms).  Synthetic model: A = random symmetric, w random -- timing/traffic only.
usage: prof_scan.py [N] [M] [digits] [reps] [what=scan|kin|both]"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mixmogam_amd import _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
D = int(sys.argv[3]) if len(sys.argv) > 3 else 4
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
what = sys.argv[5] if len(sys.argv) > 5 else "scan"
ctx = _lib.Context(0)
g = ctx.geno(M=M, N=N).fill_hash(20240)
if what in ("scan", "both"):
    rng = np.random.RandomState(0)
    B = rng.standard_normal((N, 64)) / 8.0
    A = np.eye(N) + (B @ B.T) / N
    w = rng.standard_normal(N)
    ctx.scan_set_model(A, w, D)
    for _ in range(reps):
        ctx.scan(g, 1.0e9, N - 2, fetch=False)
    print("scan_quad ms", ctx.kernel_ms("scan_quad"), "finalize ms", ctx.kernel_ms("scan_finalize"))
if what in ("kin", "both"):
    for _ in range(reps):
        ctx.kinship_ibs_counts(g)
    print("kinship_i8 ms", ctx.kernel_ms("kinship"), "transpose ms", ctx.kernel_ms("pack"))

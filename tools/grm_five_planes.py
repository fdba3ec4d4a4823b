#!/usr/bin/env python3
"""Five weight planes on a binary store (a short call, or rare variants stretching the weight range): the one-pass kernel for
planes 1-4 plus one image GEMM for plane 0 (default) against five image GEMMs (MMG_GRM_HYBRID=0), forced with MMG_GRM_PLANES=5
on the hash store.   MMG_GRM_PLANES=5 [MMG_GRM_HYBRID=0] python tools/grm_five_planes.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mixmogam_amd import _lib
ctx = _lib.get_context()
for N, M in ((5000, 1000000), (50000, 100000)):
    g = ctx.geno(M=M, N=N).fill_hash(20240)
    acc = ctx.kinship_accumulator(N)
    acc.add_grm(g)
    best = 1e9
    for _ in range(3):
        t0 = time.time(); acc.add_grm(g); dt = time.time() - t0
        if dt < best: best, kern = dt, ctx.kernel_ms("grm")
    print("binary N=%d M=%d planes=%s hybrid=%s: %.1f ms wall, GEMMs %.1f ms" % (N, M, os.environ.get("MMG_GRM_PLANES", "auto"), os.environ.get("MMG_GRM_HYBRID", "1"), best * 1e3, kern), flush=True)
    acc.close(); g.close()

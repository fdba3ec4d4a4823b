"""Exact GRM of one chunk: python tools/grm_time.py N M  -- wall time of mmg_kin_acc_add_grm and its device passes."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixmogam_amd import _lib
N, M = int(sys.argv[1]), int(sys.argv[2])
ctx = _lib.get_context()
g = ctx.geno(M=M, N=N).fill_hash(1)
acc = ctx.kinship_accumulator(N)
for rep in range(3):
    t0 = time.time(); acc.add_grm(g); dt = time.time() - t0
    print("N=%d M=%d add_grm %.3f s: digit-plane GEMMs %.1f ms, image pass %.1f ms" % (N, M, dt, ctx.kernel_ms("grm"), ctx.kernel_ms("pack")), flush=True)
if len(sys.argv) > 3:
    mean, sd = g.snp_stats()
    for rep in range(2):
        t0 = time.time(); acc.add(g, 1.0 / sd, -mean / sd); print("add (fp32 MFMA) %.3f s, kernel %.1f ms" % (time.time() - t0, ctx.kernel_ms("kinship")), flush=True)

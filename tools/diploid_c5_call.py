#!/usr/bin/env python3
"""One exact-GRM call at config 5's call shape (N = 50,000 individuals x 100,000 SNPs) on a binary store and on a 0/1/2 store
(VERDICT r3 'missing' 4: config 5 calls itself human-scale, and plink2hdf5.py writes 0/1/2).  The 0/1/2 rows are a block of
8,192 random SNPs (Hardy-Weinberg from two draws at a per-SNP frequency) uploaded 2-bit packed and repeated.
    python tools/diploid_c5_call.py [N] [M]        MMG_GRM_CENTRE=0: the five 6-bit planes of before"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixmogam_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
ctx = _lib.get_context()
rng = np.random.RandomState(5)
blk = 8192
f = rng.uniform(0.1, 0.9, blk)
codes = ((rng.random_sample((blk, N)) < f[:, None]).astype(np.int8) + (rng.random_sample((blk, N)) < f[:, None]).astype(np.int8))
packed = _lib.pack_genotypes(codes, 2)
del codes
stores = {"binary": ctx.geno(M=M, N=N).fill_structured(20240, npop=3)}
g2 = ctx.geno(M=M, N=N)
for m0 in range(0, M, blk):
    rows = min(blk, M - m0)
    g2.upload_packed(packed[:rows], 2, m0)
stores["0/1/2"] = g2
for name, g in stores.items():
    acc = ctx.kinship_accumulator(N)
    acc.add_grm(g)                                              # allocates the workspace, starts the run
    best, kern = 1e9, 0.0
    for _ in range(2):
        t0 = time.time(); acc.add_grm(g); dt = time.time() - t0
        if dt < best:
            best, kern = dt, ctx.kernel_ms("grm")
    pend = acc.pending()
    acc.close()
    print("%-7s N=%d M=%d: exact GRM call %.1f ms wall, GEMMs %.1f ms (%d SNPs in the planes after three calls)%s"
          % (name, N, M, best * 1e3, kern, pend, "   [MMG_GRM_CENTRE=0]" if os.environ.get("MMG_GRM_CENTRE") == "0" else ""), flush=True)

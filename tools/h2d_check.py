#!/usr/bin/env python3
"""PCIe-inclusive note for DESIGN.md: time to bring N x M int8 genotypes from host memory into the padded store."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixmogam_amd import _lib
ctx = _lib.Context(0)
N, M = 5000, 1000000
rng = np.random.RandomState(0)
host = rng.randint(0, 2, size=(M // 10, N)).astype(np.int8)
g = ctx.geno(M=M, N=N)
for name, buf in (("pageable", host), ("pinned", None)):
    if buf is None:
        p = ctx.pinned_empty(host.size, dtype=np.int8).reshape(host.shape)
        p[...] = host
        buf = p
    g.upload(buf, 0)
    t0 = time.perf_counter()
    for k in range(10):
        g.upload(buf, k * (M // 10))
    dt = time.perf_counter() - t0
    print("%s host buffer: %.1f ms for %.2f GB -> %.1f GB/s" % (name, dt * 1e3, host.nbytes * 10 / 1e9, host.nbytes * 10 / 1e9 / dt))

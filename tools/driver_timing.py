#!/usr/bin/env python3
"""hdf5_data.run_emmax from an on-disk container at mid sizes (what a user with a few thousand individuals runs): stage
times from the driver's own `timings`, int8 and 1-bit packed containers.   python tools/driver_timing.py [N] [M]"""
import os, shutil, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixmogam_amd import _lib, chunkstore, hdf5_data
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 400000
ctx = _lib.get_context()
rng = np.random.RandomState(0)
tmp = tempfile.mkdtemp(prefix="mmg_drv_", dir="/dev/shm")
try:
    chroms = {}
    for c in range(4):
        f = rng.uniform(0.1, 0.9, M // 4)
        chroms["chr%d" % (c + 1)] = (rng.random_sample((M // 4, N)) < f[:, None]).astype(np.int8)
    y = rng.standard_normal(N) + chroms["chr1"][5]
    for bits in (0, 1):
        path = os.path.join(tmp, "g%d" % bits)
        chunkstore.write_genotype_container(path, chroms, np.arange(N), phenotypes=y, packed_bits=bits)
        for rep in range(2):
            T = {}
            t0 = time.time()
            out = hdf5_data.run_emmax(path, None, min_maf=0.1, chunk_size=100000, ctx=ctx, timings=T)
            dt = time.time() - t0
            print("N=%d M=%d packed_bits=%d run %d: %.3f s  %s" % (N, M, bits, rep, dt, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in T.items() if k != "route"}), flush=True)
finally:
    shutil.rmtree(tmp, ignore_errors=True)

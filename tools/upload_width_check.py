#!/usr/bin/env python3
"""Host -> padded store by row width: the strided 2-D copy against the contiguous copy + pitch kernel (MMG_UPLOAD_PATH), for
individual counts that are not multiples of 16 (the bundled A. thaliana set has 199).   python tools/upload_width_check.py"""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) == 1:
    for path in ("2d", "staged", "auto"):
        env = dict(os.environ, MMG_UPLOAD_PATH=path)
        print("---- MMG_UPLOAD_PATH=%s" % path, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=False)
    sys.exit(0)
from mixmogam_amd import _lib
ctx = _lib.Context(0)
rng = np.random.RandomState(0)
seen5000 = False
for n in (199, 200, 1001, 1000, 4999, 5000, 5000, 20001, 20000):
    m = max(2000, min(250000, int(2.5e8 // n))) if n != 5000 or seen5000 else 250000
    seen5000 = seen5000 or n == 5000
    host = rng.randint(0, 2, size=(m, n)).astype(np.int8)
    g = ctx.geno(M=m, N=n)
    g.upload(host, 0)
    t0 = time.perf_counter()
    for _ in range(3):
        g.upload(host, 0)
    dt = (time.perf_counter() - t0) / 3
    back = g.download_rows([0, m // 2, m - 1])
    ok = np.array_equal(back, host[[0, m // 2, m - 1]])
    print("N=%6d M=%7d: %8.2f ms  %6.2f GB/s  %s" % (n, m, dt * 1e3, host.nbytes / 1e9 / dt, "ok" if ok else "MISMATCH"), flush=True)
    g.close()

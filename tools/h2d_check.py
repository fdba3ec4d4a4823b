#!/usr/bin/env python3
"""PCIe-inclusive note for DESIGN.md: time to bring N x M genotypes from host memory into the padded store -- int8 rows
from pageable / page-locked memory through both copy paths (MMG_UPLOAD_PATH=2d|staged, one process each), fp32 rows,
1-bit packed rows."""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) == 1:
    for path in ("auto", "staged"):
        env = dict(os.environ)
        if path != "auto":
            env["MMG_UPLOAD_PATH"] = path
        print("---- MMG_UPLOAD_PATH=%s" % path, flush=True)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, check=False)
    sys.exit(0)
from mixmogam_amd import _lib
ctx = _lib.Context(0)
N, M = 5000, 1000000
rng = np.random.RandomState(0)
host = rng.randint(0, 2, size=(M // 10, N)).astype(np.int8)
g = ctx.geno(M=M, N=N)


def timed(name, fn, nbytes):
    fn(0)
    t0 = time.perf_counter()
    for k in range(10):
        fn(k)
    dt = time.perf_counter() - t0
    print("%-34s %.1f ms for %.2f GB of host bytes -> %.1f GB/s, %.1f M SNPs/s"
          % (name, dt * 1e3, nbytes * 10 / 1e9, nbytes * 10 / 1e9 / dt, M / dt / 1e6), flush=True)


pin = ctx.pinned_empty(host.size, dtype=np.int8).reshape(host.shape)
pin[...] = host
timed("int8 pageable", lambda k: g.upload(host, k * (M // 10)), host.nbytes)
timed("int8 page-locked", lambda k: g.upload(pin, k * (M // 10)), host.nbytes)
packed = _lib.pack_genotypes(host, 1)
timed("1-bit packed pageable", lambda k: g.upload_packed(packed, 1, k * (M // 10)), packed.nbytes)
h32 = host[: M // 40].astype(np.float32)
timed("fp32 pageable (quarter)", lambda k: g.upload(h32, k * (M // 40)), h32.nbytes)

#!/usr/bin/env python3
"""Wall time of mmg_kin_acc_add_grm per call next to its GEMM kernels, with and without runs of calls sharing the digit
planes (MMG_GRM_DEFER, include/mixmogam_hip.h): C3 (N = 5000 x M = 10^6, one call) and the config-5 shape (N = 50,000,
calls of 65,536 SNPs).  Each setting runs in a process of its own (the switch is read once).
    python tools/grm_call_cost.py"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np
    sys.path.insert(0, ROOT)
    from mixmogam_amd import _lib
    ctx = _lib.get_context()
    for n, m, calls in ((5000, 1000000, 3), (50000, 65536, 6)):
        g = ctx.geno(M=m, N=n).fill_hash(20240)
        acc = ctx.kinship_accumulator(n)
        acc.add_grm(g)                                         # workspace allocation, first-call costs
        acc.fetch() if n <= 5000 else acc.scale_k()
        ctx.synchronize() if hasattr(ctx, "synchronize") else None
        t0 = time.time()
        gemm = 0.0
        for _ in range(calls):
            acc.add_grm(g)
            gemm += ctx.kernel_ms("grm")
        t1 = time.time()
        acc.scale_k()
        t2 = time.time()
        print("  N=%d M=%d: %.1f ms per call (GEMM kernels %.1f ms), reader after %d calls %.1f ms, pending before it %s"
              % (n, m, (t1 - t0) / calls * 1e3, gemm / calls, calls, (t2 - t1) * 1e3, "yes" if calls else "-"), flush=True)
        acc.close(); g.close()
    sys.exit(0)
for defer in ("1", "0"):
    print("MMG_GRM_DEFER=%s" % defer, flush=True)
    env = dict(os.environ, MMG_GRM_DEFER=defer)
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, check=True)

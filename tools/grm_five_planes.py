#!/usr/bin/env python3
"""Five weight planes on a binary store (a short call, or rare variants stretching the weight range): the one-pass kernel for
planes 1-4 plus one image GEMM for plane 0 (default) against five image GEMMs (MMG_GRM_HYBRID=0), forced with MMG_GRM_PLANES=5
on a synthetic store -- time per call and the accumulated matrix of both (every plane is an exact integer sum: they must
agree bit for bit).  The switches are read once per process, so each mode runs in a process of its own.
    python tools/grm_five_planes.py [N M [N M ...]]        default: 5000 1000000 50000 100000"""
import hashlib, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    from mixmogam_amd import _lib
    ctx = _lib.get_context()
    shapes = [int(v) for v in sys.argv[2:]]
    for N, M in zip(shapes[::2], shapes[1::2]):
        g = ctx.geno(M=M, N=N).fill_structured(20251, npop=3)
        acc = ctx.kinship_accumulator(N)
        acc.add_grm(g)
        best, kern = 1e9, 0.0
        for _ in range(3):
            t0 = time.time(); acc.add_grm(g); dt = time.time() - t0
            if dt < best:
                best, kern = dt, ctx.kernel_ms("grm")
        K, cnt = acc.fetch()
        print("  mode %-6s N=%d M=%d: %.1f ms wall, GEMMs %.1f ms; sum of 4 calls: sha1 %s" % (
            "images" if os.environ.get("MMG_GRM_HYBRID") == "0" else "hybrid", N, M, best * 1e3, kern,
            hashlib.sha1(K.tobytes()).hexdigest()[:16]), flush=True)
        acc.close(); g.close()
    sys.exit(0)
shapes = sys.argv[1:] or ["5000", "1000000", "50000", "100000"]
for mode in ("hybrid", "images"):
    env = dict(os.environ, MMG_GRM_PLANES="5")
    if mode == "images":
        env["MMG_GRM_HYBRID"] = "0"
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + shapes, env=env, check=False)

#!/usr/bin/env python3
"""The one-pass GRM kernel in its three generations (default since round 5: 2 x 2 quadrants, every slice scales its own operands;
MMG_GRM4_LAYOUT=quad: rounds 3-4, scaling written one slice ahead; =strips: four row strips): kernel time at C3 and the accumulated
matrix of each (they must agree bit for bit: every plane is an exact integer sum).
    python tools/grm4_layouts.py [N] [M]"""
import os, subprocess, sys, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np
    sys.path.insert(0, ROOT)
    from mixmogam_amd import _lib
    n, m = int(sys.argv[2]), int(sys.argv[3])
    ctx = _lib.get_context()
    g = ctx.geno(M=m, N=n).fill_structured(20250, npop=3)
    acc = ctx.kinship_accumulator(n)
    ms = []
    for _ in range(4):
        acc.add_grm(g)
        ms.append(ctx.kernel_ms("grm"))
    K, cnt = acc.fetch()
    print("  layout %-6s: %.2f ms (%s)  sum of 4 calls: sha1 %s  K[0,0] %.12g K[7,3] %.12g" % (
        os.environ.get("MMG_GRM4_LAYOUT", "jit"), min(ms[1:]), " ".join("%.2f" % x for x in ms),
        hashlib.sha1(K.tobytes()).hexdigest()[:16], K[0, 0], K[7, 3]), flush=True)
    sys.exit(0)
n = sys.argv[1] if len(sys.argv) > 1 else "5000"
m = sys.argv[2] if len(sys.argv) > 2 else "1000000"
for layout in ("jit", "quad", "strips"):
    env = dict(os.environ)
    env.pop("MMG_GRM4_LAYOUT", None)
    if layout != "jit":
        env["MMG_GRM4_LAYOUT"] = layout
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child", n, m], env=env, check=False)

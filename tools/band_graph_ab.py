#!/usr/bin/env python3
"""The band reduction of K as 1200 stream launches against one hipGraph of them (MMG_BAND_GRAPH=1): seconds of
mmg_reml_band_info for a fresh workspace, three workspaces per setting.   python tools/band_graph_ab.py [N]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import numpy as np
    sys.path.insert(0, ROOT)
    from mixmogam_amd import _lib
    n = int(sys.argv[2])
    ctx = _lib.get_context()
    rng = np.random.RandomState(0)
    B = rng.standard_normal((n, 64))
    K = B @ B.T / 64 + 0.5 * np.eye(n)
    y = rng.standard_normal(n); X = np.ones((n, 1))
    out = []
    ref = None
    for rep in range(4):
        r = ctx.reml(K, X, y)
        s = r.sums(np.array([0.5, 2.0]), route="band")
        out.append(r.band_info()["seconds"] * 1e3)
        ref = s if ref is None else ref
        assert all(np.array_equal(a, b) for a, b in zip(s[:4], ref[:4]))
        r.close()
    print("  MMG_BAND_GRAPH=%s N=%d: band reduction %s ms; s1 %.12g" % (os.environ.get("MMG_BAND_GRAPH", "0"), n, " ".join("%.2f" % x for x in out), ref[0][0]), flush=True)
    sys.exit(0)
n = sys.argv[1] if len(sys.argv) > 1 else "5000"
for gsel in ("0", "1"):
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child", n], env=dict(os.environ, MMG_BAND_GRAPH=gsel, MMG_BAND_GRAPH_VERBOSE="1"), check=False)

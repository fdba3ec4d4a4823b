#!/usr/bin/env python3
"""Quick parity + timing check of the scan kernel variant selected by MMG_SCAN_KERNEL (GPU box)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_case
from oracle import emmax_oracle as orc
from mixmogam_amd import _lib
ctx = _lib.Context(0)
worst = 0.0
for name in ("struct_n150_s0", "struct_n300_s3"):
    case = load_case(name)
    y = case["y"]; n = len(y)
    X = np.ones((n, 1))
    if case["cof"] is not None:
        X = np.hstack([X] + [c.reshape(n, 1) for c in case["cof"]])
    est = orc.get_estimates(y, X, orc.scale_k(case["dbl_ibs_scaled"]))
    prep = orc.scan_prepare(y, X, est["H_sqrt_inv"])
    ref = orc.scan_closed(case["snps"], prep)
    ctx.scan_set_model(prep["A"], prep["w"], 4)
    out = ctx.scan(ctx.geno(case["snps"]), prep["h0_rss"], n - X.shape[1] - 1)
    worst = max(worst, float(np.max(np.abs(out["ps"] / ref["ps"] - 1))))
# mid size with several column tiles, vs the default kernel bit for bit
rng = np.random.RandomState(1)
n, m = 1100, 5000
snps = (rng.random_sample((m, n)) < rng.uniform(0.05, 0.95, size=(m, 1))).astype(np.int8)
B = rng.standard_normal((n, 40)) / 6
A = np.eye(n) + B @ B.T / n
w = rng.standard_normal(n)
ctx.scan_set_model(A, w, 4)
g = ctx.geno(snps)
var = os.environ.pop("MMG_SCAN_KERNEL", None)
abl = os.environ.pop("MMG_ABLATE", None)
base = ctx.scan(g, 1e6, n - 2, stats=True)
if var:
    os.environ["MMG_SCAN_KERNEL"] = var
if abl:
    os.environ["MMG_ABLATE"] = abl
alt = ctx.scan(g, 1e6, n - 2, stats=True)
same = all(np.array_equal(base[k], alt[k]) for k in ("rss", "den", "ps"))
S = snps.astype(float)
den = np.einsum("ij,ij->i", S @ A, S)
print("variant", var, "max rel p err golden %.2e" % worst, "bitwise == default:", same,
      "den rel err %.2e" % float(np.max(np.abs(alt["den"] / den - 1))))

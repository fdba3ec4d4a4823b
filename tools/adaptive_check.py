#!/usr/bin/env python3
"""Accuracy of the adaptive digit schedule on the golden cases (GPU box): max relative p error of the default model
vs the oracle, the share of refined SNPs and the error-model statistics, beside the explicit 4-plane scan."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_case
from oracle import emmax_oracle as orc
from mixmogam_amd import _lib
ctx = _lib.Context(0)
for name in ("struct_n150_s0", "struct_n150_s1", "struct_n300_s2", "struct_n300_s3", "bern_n200_s4"):
    case = load_case(name)
    y = case["y"]; n = len(y)
    X = np.ones((n, 1))
    if case["cof"] is not None:
        X = np.hstack([X] + [c.reshape(n, 1) for c in case["cof"]])
    est = orc.get_estimates(y, X, orc.scale_k(case["dbl_ibs_scaled"]))
    prep = orc.scan_prepare(y, X, est["H_sqrt_inv"])
    ref = orc.scan_closed(case["snps"], prep)
    g = ctx.geno(case["snps"])
    out = {}
    for nd in (4, 0):
        ctx.scan_set_model(prep["A"], prep["w"], nd)
        r = ctx.scan(g, prep["h0_rss"], n - X.shape[1] - 1)
        out[nd] = (float(np.max(np.abs(r["ps"] / ref["ps"] - 1))), ctx.scan_last_stats())
    print(name, "M=%d" % len(ref["ps"]), "4 planes: %.2e" % out[4][0], "| adaptive: %.2e" % out[0][0], out[0][1])

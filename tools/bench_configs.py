#!/usr/bin/env python3
"""Secondary measurements for BASELINE.json configs[1] (C2) and configs[3] (C4, permutations) on 1 GPU.
Prints one JSON line per config.  usage: bench_configs.py [c2] [c4] [--perms P] [--m M]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mixmogam_amd import _lib, kinship, linear_models as lm


def setup(ctx, N, M, seed=20240):
    g = ctx.geno(M=M, N=N).fill_hash(seed)
    rng = np.random.RandomState(20241)
    rows = g.download(0, 100).astype(np.float64)
    gen = rng.exponential(1.0, size=100) @ rows
    err = rng.normal(0, 1, size=N)
    y = gen + err * np.sqrt(0.25 * np.var(gen, ddof=1) / np.var(err, ddof=1))
    y = (y - y.mean()) / y.std()
    t0 = time.time()
    counts = ctx.kinship_ibs_counts(g)
    kin_ms = ctx.kernel_ms("kinship")
    K = kinship.scale_k(counts / (2.0 * M) + 0.5)
    lmm = lm.LinearMixedModel(y, ctx=ctx)
    lmm.add_random_effect(K)
    eig_L = lmm._get_eigen_L_()
    eig_R = lmm._get_eigen_R_(X=lmm.X)
    est = lmm._get_estimates_with(eig_L, eig_R, "REML")
    return g, lmm, est, kin_ms, time.time() - t0


def c2(ctx, N=1000, M=500000):
    g, lmm, est, kin_ms, setup_s = setup(ctx, N, M)
    prep = lmm.scan_prepare(est["H_sqrt_inv"])
    ctx.scan_set_model(prep["A"], prep["w"], 4)
    bufs = tuple(ctx.pinned_empty(M) for _ in range(3))

    def one():
        ctx._check(ctx.lib.mmg_emmax_scan_device(ctx.h, g.h, float(prep["h0_rss"]), int(prep["n_p"])))
        ctx._check(ctx.lib.mmg_scan_fetch(ctx.h, M, *[_lib._ptr(b) for b in bufs]))
    one()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        one()
    dt = (time.perf_counter() - t0) / reps
    out = {"ps": bufs[2]}
    return {"config": "C2 simulations.py synthetic N=%d x M=%d, IBS kinship + EMMAX scan, 1 GPU" % (N, M),
            "snps_per_s": M / dt, "ms_per_scan": dt * 1e3, "scan_quad_ms": ctx.kernel_ms("scan_quad"),
            "finalize_ms": ctx.kernel_ms("scan_finalize"), "kinship_i8_ms": kin_ms,
            "kinship_algorithmic_tops": 2.0 * N * N * M / (kin_ms * 1e-3) / 1e12, "setup_s": setup_s,
            "min_p": float(out["ps"].min())}


def c4(ctx, N=5000, M=1000000, P=1000):
    g, lmm, est, kin_ms, setup_s = setup(ctx, N, M)
    rng = np.random.RandomState(20242)
    perm_idx = np.array([rng.permutation(N) for _ in range(P)])
    t0 = time.time()
    res = lmm._emmax_permutations_(g, None, est["H_sqrt_inv"], num_perm=P, perm_idx=perm_idx)
    dt = time.time() - t0
    pm = ctx.kernel_ms("perm")
    return {"config": "C4 EMMAX permutation test N=%d x M=%d, P=%d, 1 GPU" % (N, M, P),
            "wall_s": dt, "perm_gemm_ms": pm, "perm_gemm_algorithmic_tflops": 2.0 * N * P * M / (pm * 1e-3) / 1e12,
            "perm_snp_perm_pairs_per_s": M * P / dt, "threshold_05_min_p": float(np.sort(res["min_ps"])[P // 20]),
            "setup_s": setup_s}


if __name__ == "__main__":
    ctx = _lib.Context(0)
    args = sys.argv[1:]
    P = int(args[args.index("--perms") + 1]) if "--perms" in args else 1000
    M = int(args[args.index("--m") + 1]) if "--m" in args else None
    if "c2" in args or not [a for a in args if a in ("c2", "c4")]:
        print(json.dumps(c2(ctx, M=M or 500000)))
    if "c4" in args or not [a for a in args if a in ("c2", "c4")]:
        print(json.dumps(c4(ctx, M=M or 1000000, P=P)))

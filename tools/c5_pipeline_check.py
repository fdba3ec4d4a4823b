#!/usr/bin/env python3
"""C5-shaped (large N) end-to-end check of the whole path on one GPU: hash genotypes -> exact IBS
kinship -> device eigh -> REML -> scan model (device dgemm) -> EMMAX scan, with stage timings and
sample checks against float64 host arithmetic.
usage: c5_pipeline_check.py N M"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mixmogam_amd import _lib, kinship, linear_models as lm
from oracle import emmax_oracle as orc

N = int(sys.argv[1]); M = int(sys.argv[2])
ctx = _lib.Context(0)
print("host mem:", os.popen("free -g | sed -n 2p").read().strip(), flush=True)
T = {}


def timed(name, fn):
    t0 = time.time()
    r = fn()
    T[name] = round(time.time() - t0, 2)
    print("%-14s %.2f s" % (name, T[name]), flush=True)
    return r


g = ctx.geno(M=M, N=N).fill_hash(20240)
counts = timed("kinship", lambda: ctx.kinship_ibs_counts(g))
T["kinship_kernel_ms"] = ctx.kernel_ms("kinship")
assert np.all(np.diag(counts) == M)
cols = np.r_[0:24, N - 24:N]
sub = orc.hash_genotypes(0, M, N, 20240)[:, cols] if N * M <= 4e9 else g.download()[:, cols]
x = 2 * sub.astype(np.int64) - 1
assert np.array_equal(counts[np.ix_(cols, cols)], (x.T @ x).astype(counts.dtype)), "kinship corner mismatch"
assert np.array_equal(counts[cols, :][:, ::997], counts[::997, :][:, cols].T)
K = timed("scale_k", lambda: kinship.scale_k(counts / (2.0 * M) + 0.5))
del counts

rng = np.random.RandomState(1)
causal = rng.choice(M, 10, replace=False)
gen = rng.exponential(1.0, 10) @ np.vstack([g.download(int(c), 1)[0] for c in causal]).astype(np.float64)
err = rng.standard_normal(N)
y = gen + err * np.sqrt(0.25 * gen.var() / err.var())
y = (y - y.mean()) / y.std()

lmm = lm.LinearMixedModel(y, ctx=ctx)
lmm.add_random_effect(K)
eig_L = timed("eigh_L", lambda: lmm._get_eigen_L_())
T["eigh_kernel_ms"] = ctx.kernel_ms("eigh")
vals, vecs = eig_L["values"], eig_L["vectors"]
assert np.all(np.diff(vals) >= -1e-9 * abs(vals).max())
Ks = lmm.random_effects[1][1] if hasattr(lmm, "random_effects") else K
for i in (0, N // 2, N - 1):                                   # rows are eigenvectors
    v = np.asarray(vecs[i]).reshape(-1)
    r = np.asarray(Ks @ v).reshape(-1) - vals[i] * v
    assert np.linalg.norm(r) < 1e-8 * max(1.0, abs(vals).max()), ("eig residual", i, np.linalg.norm(r))
est = timed("reml", lambda: lmm.get_estimates(eig_L, method="REML"))   # sums from eig_L alone: no second eigh
prep = timed("scan_prepare", lambda: lmm.scan_prepare(est["H_sqrt_inv"]))
timed("set_model", lambda: ctx.scan_set_model(prep["A"], prep["w"], 0))
out = timed("scan", lambda: ctx.scan(g, prep["h0_rss"], prep["n_p"], stats=True))
T["scan_quad_ms"] = ctx.kernel_ms("scan_quad"); T["scan_finalize_ms"] = ctx.kernel_ms("scan_finalize")
T["adaptive_scan"] = ctx.scan_last_stats()

idx = np.unique(np.r_[np.argsort(out["ps"])[:8], np.linspace(0, M - 1, 16).astype(int)])
worst = 0.0
H = np.asarray(est["H_sqrt_inv"])
hX = H @ lmm.X
Q, _ = np.linalg.qr(hX)
for i in idx:                                                   # float64 host evaluation of linear_models.py:1316-1349
    s = g.download(int(i), 1)[0].astype(np.float64)
    t = H @ s
    t = t - Q @ (Q.T @ t)
    rss = prep["h0_rss"] - float(t @ prep["r"]) ** 2 / float(t @ t)
    F = (prep["h0_rss"] / rss - 1) * prep["n_p"]
    p = float(orc.f_sf(np.array([F]), 1, prep["n_p"])[0])
    if p > 1e-290:
        worst = max(worst, abs(out["ps"][i] / p - 1))
print({"N": N, "M": M, "delta": float(est["delta"]), "pseudo_h2": float(est["pseudo_heritability"]),
       "min_p": float(out["ps"].min()), "max_rel_p_err_vs_host_f64": worst, "timings": T})
assert worst < 1e-6

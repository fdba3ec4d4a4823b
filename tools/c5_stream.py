#!/usr/bin/env python3
"""BASELINE config 5 -- N = 50,000 x M = 10,000,000 streamed from chunks over 8 GPUs -- as ONE rank of the 8 sees it
(or the whole thing with --world 1): the rank's SNP share is written chunk-wise to an on-disk container (never fully
in host memory), then hdf5_data.run_emmax streams it twice (GRM kinship pass, scan pass) through the double-buffered
H2D path.  Prints stage timings, SNPs/s including ingest, and checks sampled p-values against float64 host arithmetic.

    python tools/c5_stream.py [--n 50000] [--m-total 10000000] [--world 8] [--chunk 50000] [--scratch /dev/shm]
"""
import argparse
import json
import os
import shutil
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mixmogam_amd import _lib, chunkstore, hdf5_data, linear_models as lm, simulations  # noqa: E402
from oracle import emmax_oracle as orc  # noqa: E402  (checker only)

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=50000)
ap.add_argument("--m-total", type=int, default=10000000)
ap.add_argument("--world", type=int, default=8, help="ranks the SNP axis is split over; this run plays rank 0")
ap.add_argument("--chunk", type=int, default=50000)
ap.add_argument("--scratch", default="/dev/shm")
ap.add_argument("--keep", action="store_true")
ap.add_argument("--lazy", action="store_true",
                help="no container: every chunk is regenerated on each read (simulations.LazySyntheticGenotypes), "
                     "nothing but the current chunks is ever in host memory -- the whole 10 M SNPs on one GPU")
ap.add_argument("--packed", action="store_true",
                help="with --lazy: the same genotypes as 1-bit packed rows (raw_snps_packed), expanded on the device")
ap.add_argument("--packed-bits", type=int, default=1, choices=(1, 2),
                help="with --packed: 1 = 0/1 genotypes, one bit each; 2 = 0/1/2 codes (plink2hdf5.py:171-179), two bits each")
ap.add_argument("--max-gb", type=float, default=80.0, help="largest container this run may put into the scratch directory")
ap.add_argument("--writers", type=int, default=8, help="processes generating / writing the container")
ap.add_argument("--eig", action="store_true", help="take the eigendecomposition route (eigh of K) even beyond N = 46,340")
ap.add_argument("--samples", type=int, default=12, help="SNPs whose p-values are recomputed in float64 on the host (4 top hits + random)")
a = ap.parse_args()
N, M, CH = a.n, a.m_total // a.world, a.chunk
root = os.path.join(a.scratch, "mmg_c5_%d" % os.getpid())
need = 0 if a.lazy else N * M
free = shutil.disk_usage(a.scratch).free
print("share of rank 0: N=%d x M=%d = %.1f GB %s" % (N, M, N * M / 1e9, "regenerated on every read, never stored" if a.lazy
      else "container in %s (%.0f GB free)" % (a.scratch, free / 1e9)), flush=True)
assert free > 1.2 * need, "not enough scratch space"
# A container in /dev/shm is host MEMORY, and a GPU box's job is limited far below what `free` shows for the machine:
# the full 10 M-SNP container (500 GB) took a box down in round 2.  Refuse anything beyond --max-gb.
assert need <= a.max_gb * 1e9, "container of %.0f GB exceeds --max-gb %.0f (host memory of the job)" % (need / 1e9, a.max_gb)
ctx = _lib.Context(0)
T = {}
# heartbeat: the eigendecomposition at N = 50,000 is silent for several minutes, and a GPU box takes a job that
# writes nothing for 7 minutes to be hung
import threading
_t00 = time.time()
_stop = threading.Event()


def _beat():
    while not _stop.wait(60.0):
        print("[%4.0f s] working ..." % (time.time() - _t00), flush=True)


threading.Thread(target=_beat, daemon=True).start()
try:
    GEN = 6250                                               # generation chunk of the lazy source (8 per 50,000-SNP read)
    if a.lazy:
        tree, y_lazy = simulations.lazy_synthetic_source(N, M, num_chroms=5, gen_rows=GEN, seed=20240, pheno_seed=20241,
                                                         num_causals=100, threads=a.writers,
                                                         packed=a.packed_bits if a.packed else 0)
    t0 = time.time()
    path = None if a.lazy else simulations.write_synthetic_container(os.path.join(root, "geno.mmg"), N, M, chunk_rows=CH, num_chroms=5,
                                                 seed=20240, pheno_seed=20241, num_causals=100, workers=a.writers)
    T["write_container_s"] = round(time.time() - t0, 1)
    if not a.lazy:
        print("container written: %.1f s (%.2f GB/s)" % (T["write_container_s"], need / 1e9 / max(T["write_container_s"], 1e-9)), flush=True)
    src = {"genot_data": tree, "phenotypes": y_lazy} if a.lazy else hdf5_data.open_hdf5(path)

    def regenerate(gi):                                       # genotype row of global SNP index gi, from the generator
        if not a.lazy:
            cid, off = divmod(int(gi), CH)
            return simulations.synthetic_chunk(cid, CH, N, 20240)[off]
        for chrom in tree:
            ds = tree[chrom]["raw_snps_packed" if a.packed else "raw_snps"]
            if gi < len(ds):
                return _lib.unpack_genotypes(ds[int(gi)][None, :], N, a.packed_bits)[0] if a.packed else ds[int(gi)]
            gi -= len(ds)
        raise IndexError(gi)
    plan = hdf5_data._chunk_plan(src["genot_data"], 0.1, CH)
    t0 = time.time()
    K, n_snps = hdf5_data._ibd_kinship(ctx, src["genot_data"], N, plan)
    T["kinship_pass_s"] = round(time.time() - t0, 1)
    print("kinship pass (ingest + exact int8 GRM, %d chunks): %.1f s = %.2f M SNPs/s, %.1f GB/s ingest"
          % (len(plan), T["kinship_pass_s"], M / T["kinship_pass_s"] / 1e6, N * M / 1e9 / T["kinship_pass_s"]), flush=True)
    y = src["phenotypes"]
    lmm = lm.LinearMixedModel(y, ctx=ctx)
    lmm.add_random_effect(K)
    eigen_free = N > 46340 and not a.eig
    _band = False
    if eigen_free:
        # REML from Cholesky factorisations of K + delta I, scan model P(delta), P y built in HBM: no eigh at all
        t0 = time.time()
        est = lmm.get_estimates_eigen_free()
        _band = est["reml"].uses_band()
        T["reml_cholesky_s"] = round(time.time() - t0, 1)
        print("REML, eigendecomposition-free (%s, %d variance ratios evaluated): %.1f s (pseudo-h2 %.4f)"
              % ("one band reduction of K" if _band else "a Cholesky factorisation + triangular inverse each",
                 est["n_factorisations"], T["reml_cholesky_s"], est["pseudo_heritability"]), flush=True)
        t0 = time.time()
        prep = lmm.scan_model_eigen_free(est)
        est.pop("reml").close()
        T["scan_model_s"] = round(time.time() - t0, 1)
        print("scan model P(delta), P y on the device: %.1f s" % T["scan_model_s"], flush=True)
    else:
        t0 = time.time()
        eig_L = lmm._get_eigen_L_()
        T["eigh_s"] = round(time.time() - t0, 1)
        print("eigh: %.1f s" % T["eigh_s"], flush=True)
        t0 = time.time()
        est = lmm.get_estimates(eig_L, method="REML")
        prep = lmm.scan_prepare(est["H_sqrt_inv"])
        ctx.scan_set_model(prep["A"], prep["w"], 0)
        T["reml_and_model_s"] = round(time.time() - t0, 1)
        print("REML + scan model: %.1f s (pseudo-h2 %.4f)" % (T["reml_and_model_s"], est["pseudo_heritability"]), flush=True)
    t0 = time.time()
    ps = np.empty(M)
    at = 0
    quad_ms = 0.0
    for ci, chrom, g in hdf5_data._resident_chunks(ctx, src["genot_data"], plan, reuse=True):
        p = ctx.scan(g, prep["h0_rss"], prep["n_p"])["ps"]
        quad_ms += ctx.kernel_ms("scan_quad")
        g.close()
        ps[at:at + len(p)] = p
        at += len(p)
    T["scan_pass_s"] = round(time.time() - t0, 2)
    T["scan_quad_kernel_s"] = round(quad_ms / 1e3, 2)
    print("scan pass (ingest overlapped with the scan): %.2f s = %.3f M SNPs/s incl. ingest (GEMM kernels %.2f s)"
          % (T["scan_pass_s"], M / T["scan_pass_s"] / 1e6, quad_ms / 1e3), flush=True)
    # ---- sample check in float64 on the host, independent of either route: with H = K + delta I the reference's
    # per-SNP statistic (linear_models.py:1316-1349) is rss = y'Py - (s'Py)^2 / (s'Ps), P = H^-1 - H^-1X(X'H^-1X)^-1X'H^-1;
    # the H^-1 b are solved by conjugate gradients on the dense kinship (no factorisation, no eigenvectors)
    Ks = lmm.random_effects[1][1]
    delta = float(est["delta"])

    def solve(b):
        x = np.zeros_like(b)
        r = b.copy()
        p = r.copy()
        rs = float(r @ r)
        b2 = float(b @ b)
        for _ in range(500):
            Ap = Ks @ p + delta * p
            alpha = rs / float(p @ Ap)
            x += alpha * p
            r -= alpha * Ap
            rs_new = float(r @ r)
            if rs_new < 1e-26 * b2:
                break
            p = r + (rs_new / rs) * p
            rs = rs_new
        return x

    X = lmm.X
    HiX = np.column_stack([solve(np.ascontiguousarray(X[:, c])) for c in range(X.shape[1])])
    Hiy = solve(np.asarray(y, dtype=np.float64))
    a_ = X.T @ HiX
    Py = Hiy - HiX @ np.linalg.solve(a_, X.T @ Hiy)
    h0_rss = float(np.asarray(y) @ Py)
    print("h0_rss device %.12g vs host CG %.12g" % (prep["h0_rss"], h0_rss), flush=True)
    rng = np.random.RandomState(3)
    hits = np.argsort(ps)
    hits = hits[ps[hits] > 1e-280][:4]
    sample = np.unique(np.r_[hits, rng.choice(M, max(a.samples - len(hits), 1), replace=False)])
    t_check = time.time()
    worst = 0.0
    for gi in sample:
        s = regenerate(gi).astype(np.float64)
        His = solve(s)
        den = float(s @ His) - float((X.T @ His) @ np.linalg.solve(a_, X.T @ His))
        rss = h0_rss - float(s @ Py) ** 2 / den
        F = (h0_rss / rss - 1) * prep["n_p"]
        p = float(orc.f_sf(np.array([F]), 1, prep["n_p"])[0])
        if p > 1e-290:
            worst = max(worst, abs(ps[gi] / p - 1))
    total = sum(v for k, v in T.items() if k.endswith("_s") and k not in ("write_container_s", "scan_quad_kernel_s"))
    print(json.dumps({"config": ("C5 share of rank 0 of %d" % a.world) + (", lazily generated" if a.lazy else ""), "N": N, "M_share": M, "M_total": a.m_total,
                      "codes": "0/1/2, 2-bit rows" if a.packed and a.packed_bits == 2 else "0/1",
                      "chunks": len(plan), "timings": T, "pipeline_s": round(total, 1),
                      "snps_per_s_end_to_end": M / total, "min_p": float(ps.min()),
                      "route": "eigendecomposition-free (REML through %s)" % ("one band reduction" if eigen_free and _band else "Cholesky factorisations") if eigen_free else "eigh",
                      "max_rel_p_err_vs_host_f64": worst, "n_sampled": int(len(sample)),
                      "h0_rss_rel_err_vs_host_f64": abs(prep["h0_rss"] / h0_rss - 1), "delta": delta,
                      "host_check_s": round(time.time() - t_check, 1), "adaptive_last_chunk": ctx.scan_last_stats()}))
    assert worst < 1e-6
    assert abs(prep["h0_rss"] / h0_rss - 1) < 1e-9
finally:
    if not a.keep:
        shutil.rmtree(root, ignore_errors=True)

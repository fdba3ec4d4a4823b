#!/usr/bin/env python
"""bench.py -- SNPs/sec of the EMMAX scan (BASELINE.json metric) on MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path -- the EMMAX scan (linear_models.py:1316-1349: per-SNP
quadratic form, F statistic, p-value) -- over this rank's batch of M synthetic SNPs that are
already resident in HBM, ending with rss/F/p for every SNP on the host of every rank
(N>1: after the RCCL all-gather).  Workload at N=1: BASELINE.json configs[2], N=5000 x
M=1,000,000 Bernoulli(0.5) genotypes (simulations.py:21-23 restated with a counter hash,
generated on the device).  N>1: weak scaling, every rank scans its own M SNPs of a G*M-SNP
data set; the kinship is built from all G*M SNPs (partial counts + RCCL all-reduce), the
eigendecomposition + REML are replicated.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

QUAD_KERNEL = "scan_quad_w4s_kernel"   # the dominant kernel (k_scan_w4s.hip); its name in rocprofv3 / profiles/
I8_MFMA_PEAK_TOPS = 5000.0     # dense int8 MFMA: 2x the ~2.5 PF bf16 rate (MI355X_MICROARCH.md, Matrix cores)
# What a bare v_mfma_i32_32x32x32_i8 loop on random operands sustains on this chip (power-limited clock 2.07 GHz;
# tools/probes/mfma_shape_probe.hip).  Reported beside the nominal peak, never instead of it.
I8_MFMA_SUSTAINED_TOPS = 4230.0
F32_MFMA_PEAK_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32 (same guide)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=5000, help="individuals")
    ap.add_argument("--m", type=int, default=1000000, help="SNPs per GPU")
    ap.add_argument("--digits", type=int, default=0,
                    help="digit planes of the scan model; 0 = the library default (4 planes, adaptive schedule)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-f32-kinship", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0,
                    help="SNPs in the CPU baseline sample (0 = 40 chunks of N, ~15-20 s of CPU work at N=5000)")
    return ap.parse_args()


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    os.environ.setdefault("MMG_DEVICE", str(local_rank))

    from mixmogam_amd import _lib, dist as mdist, kinship, linear_models as lm

    # No torch in this process: libmixmogam_hip links the system ROCm runtime and importing
    # torch's bundled one beside it segfaults.  Rendezvous = rank 0's ncclUniqueId through a
    # /tmp file keyed by the launcher's MASTER_PORT; barriers and the max-over-ranks time go
    # over RCCL itself.
    ctx = _lib.Context(local_rank)
    info = ctx.device_info()
    coll = None
    if world > 1 or os.environ.get("MMG_BENCH_FORCE_COLL"):   # the env knob exercises the RCCL path on 1 GPU
        boot = mdist.file_bootstrap(rank, world)
        coll = mdist.RcclCollectives(ctx, rank, world, boot)

    N, M, D = args.n, args.m, args.digits
    Mtot = M * world
    t_setup = time.time()

    # ---- synthetic genotypes, generated in HBM (this rank's block of the global SNP axis)
    g = ctx.geno(M=M, N=N)
    g.fill_hash(20240, m_global0=rank * M, thr16=32768)

    # ---- phenotype (simulations.py:64-85 restated): 100 causal SNPs of the global data set
    rng = np.random.RandomState(20241)
    causal = np.sort(rng.choice(Mtot, 100, replace=False))
    effects = rng.exponential(1.0, size=100)
    from_hash = _device_rows(ctx, causal, N, 20240)
    gen = effects @ from_hash.astype(np.float64)
    err = rng.normal(0, 1, size=N)
    y = gen + err * np.sqrt((0.2 / 0.8) * (np.var(gen, ddof=1) / np.var(err, ddof=1)))
    y = (y - y.mean()) / y.std()

    # ---- kinship: exact IBS counts on the int8 matrix cores (+ fp32-MFMA twin for the TFLOP/s figure)
    counts = ctx.kinship_ibs_counts(g)
    kin_i8_ms = ctx.kernel_ms("kinship")
    kin_f32_ms = None
    if not args.no_f32_kinship:
        cf = ctx.kinship_affine(g)
        kin_f32_ms = ctx.kernel_ms("kinship")
        if not np.array_equal(cf, counts.astype(np.float64)):
            raise SystemExit("fp32-MFMA and int8-MFMA kinship counts differ")
        del cf
    if coll is not None:
        counts = mdist.sharded_ibs_counts(counts, coll)
    K = kinship.scale_k(counts.astype(np.float64) / (2.0 * Mtot) + 0.5)

    # ---- eigh + REML (replicated), model -> device
    lmm = lm.LinearMixedModel(y, ctx=ctx)
    lmm.add_random_effect(K)
    t0 = time.time()
    eig_L = lmm._get_eigen_L_()
    eigh_ms = ctx.kernel_ms("eigh")
    est = lmm.get_estimates(eig_L, method="REML")       # REML sums from eig_L alone: no second eigh
    prep = lmm.scan_prepare(est["H_sqrt_inv"])
    ctx.scan_set_model(prep["A"], prep["w"], D)
    model_s = time.time() - t0
    n_p = prep["n_p"]
    t_setup = time.time() - t_setup

    def barrier():
        # every library call above blocks on its HIP stream (hipStreamSynchronize), so the device
        # is idle here; across ranks: an RCCL all-reduce.
        if coll is not None:
            coll.barrier()

    # Result buffers are allocated once and page-locked (mmg_host_alloc).  Delivery is double buffered: the
    # (rss, F, p) of step i are snapshotted on the device and all-gathered over RCCL / downloaded on a second
    # HIP stream (mmg_scan_deliver_begin) while step i+1 scans; the last delivery is awaited inside the timed
    # region, so every step's results are on the host of every rank before the closing barrier.
    outs2 = [[ctx.pinned_empty(M * world) for _ in range(3)] for _ in range(2)]
    comm_h = coll.h if coll is not None else None

    def step(i):
        ctx.scan(g, prep["h0_rss"], n_p, fetch=False)           # blocks until the kernels finish
        ctx.scan_deliver_begin(outs2[i & 1], count=M, comm=comm_h)
        return outs2[i & 1]

    for i in range(args.warmup):
        step(i)
    ctx.scan_deliver_wait()
    quad_ms, fin_ms = [], []
    barrier()
    t0 = time.time()
    for i in range(args.steps):
        out = step(i)
        quad_ms.append(ctx.kernel_ms("scan_quad"))
        fin_ms.append(ctx.kernel_ms("scan_finalize"))
    ctx.scan_deliver_wait()
    barrier()
    elapsed = time.time() - t0
    if coll is not None:
        elapsed = float(coll.allreduce(np.array([elapsed]), "max")[0])

    ps = out[2].copy()
    scan_stats = ctx.scan_last_stats()
    # For the record, outside the timed region: the same scan with all four digit planes for every SNP (what the
    # adaptive schedule is measured against) -- its time and how far the two sets of p-values are apart.
    all_planes = None
    if scan_stats["adaptive"] and rank == 0:
        ctx.scan_set_model(prep["A"], prep["w"], 4)
        ctx.scan(g, prep["h0_rss"], n_p, fetch=False)
        t1 = time.time()
        for _ in range(2):
            ctx.scan(g, prep["h0_rss"], n_p, fetch=False)
        dt_all = (time.time() - t1) / 2
        ref_out = ctx.scan(g, prep["h0_rss"], n_p)
        mine = ps[rank * M:(rank + 1) * M]
        ok = ref_out["ps"] > 1e-290
        all_planes = {"ms_per_scan_kernels_only": 1e3 * dt_all, "scan_quad_ms": ctx.kernel_ms("scan_quad"),
                      "max_rel_p_diff_adaptive_vs_all_planes": float(np.max(np.abs(mine[ok] / ref_out["ps"][ok] - 1)))}
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = Mtot * args.steps / elapsed
        qms = float(np.mean(quad_ms))
        alg_flop = (2.0 * N * N + 4.0 * N) * M                 # SURVEY 8d per-SNP figure x SNPs per launch
        achieved = alg_flop / (qms * 1e-3) / 1e12
        Npad = -(-N // 256) * 256
        nJ = Npad // 256
        Dn = D if D else 4
        plane_ops = 2.0 * 256.0 * 256.0 * 256.0 * (nJ * (nJ + 1) / 2)          # one digit plane over one 256-SNP block
        if scan_stats["adaptive"] and not scan_stats["fell_back"]:            # 3 planes for all + 1 for the refined
            exec_ops = plane_ops * ((Dn - 1) * (-(-M // 256)) + (-(-scan_stats["n_refined"] // 256)))
        else:
            exec_ops = plane_ops * Dn * (-(-M // 256)) * (2 if scan_stats["fell_back"] else 1)
        kin_exec = 2.0 * 256.0 * 256.0 * (nJ * (nJ + 1) / 2) * M     # lower-triangle tiles x contraction length
        traffic = None
        try:   # HBM bytes per launch measured with rocprofv3 PMC passes of this same command (profiles/)
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_c3.json")))
            if tj["config"] == {"n": N, "m": M, "digits": D} and tj.get("adaptive", False) == scan_stats["adaptive"]:
                traffic = tj["kernels"][QUAD_KERNEL]["hbm_bytes_corrected"]
        except Exception:
            pass
        res = {
            "metric": "SNPs/sec EMMAX scan", "value": value, "unit": "SNPs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "i8", "data": "synthetic",
            "config": {"workload": "EMMAX scan N=%d individuals x M=%d SNPs per GPU (BASELINE configs[2] shape), "
                                   "Bernoulli(0.5) hash genotypes resident in HBM, q=1" % (N, M),
                       "n_individuals": N, "snps_per_gpu": M, "snps_total": Mtot, "digits": Dn,
                       "digit_schedule": "adaptive (3 planes for every SNP, the 4th where p could move by 2.5e-7)"
                                         if scan_stats["adaptive"] else "all planes for every SNP",
                       "parallelism": "snp-block x%d" % world,
                       "delivery": "double buffered: step i's results are gathered (RCCL) / downloaded on a second "
                                   "stream while step i+1 scans; the last one is awaited inside the timed region"},
            "roofline": {"bound": "mfma", "kernel": QUAD_KERNEL, "achieved": achieved,
                         "peak": I8_MFMA_PEAK_TOPS, "unit": "TFLOP/s", "frac": achieved / I8_MFMA_PEAK_TOPS,
                         "traffic": traffic, "traffic_unit": "bytes per launch (PMC, profiles/traffic_c3.json)",
                         "algorithmic_bytes": float(-(-M // 256) * 256 * Npad + Dn * Npad * Npad), "ms": qms,
                         "executed_int8_tops": exec_ops / (qms * 1e-3) / 1e12,
                         "executed_frac": exec_ops / (qms * 1e-3) / 1e12 / I8_MFMA_PEAK_TOPS,
                         "executed_frac_of_sustained_mfma_rate": exec_ops / (qms * 1e-3) / 1e12 / I8_MFMA_SUSTAINED_TOPS},
            "finalize_kernel": {"ms": float(np.mean(fin_ms)),
                                "hbm_gbps": (M * (Npad + 56.0)) / (np.mean(fin_ms) * 1e-3) / 1e9},
            # "flop" is the full product the reference forms; only the lower triangle of 256^2
            # tiles is executed (kin_exec), so the *_frac_of_peak figures can exceed 1 -- the executed ones cannot
            "kinship": {"flop": 2.0 * N * N * M, "executed_flop": kin_exec,
                        "i8_executed_tops": kin_exec / (kin_i8_ms * 1e-3) / 1e12,
                        "i8_executed_frac_of_peak": kin_exec / (kin_i8_ms * 1e-3) / 1e12 / I8_MFMA_PEAK_TOPS,
                        "f32_executed_tflops": None if kin_f32_ms is None else kin_exec / (kin_f32_ms * 1e-3) / 1e12,
                        "f32_executed_frac_of_peak": None if kin_f32_ms is None else
                        kin_exec / (kin_f32_ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS,
                        "i8_ms": kin_i8_ms, "i8_tops": 2.0 * N * N * M / (kin_i8_ms * 1e-3) / 1e12,
                        "i8_frac_of_peak": 2.0 * N * N * M / (kin_i8_ms * 1e-3) / 1e12 / I8_MFMA_PEAK_TOPS,
                        "f32_ms": kin_f32_ms,
                        "f32_tflops": None if kin_f32_ms is None else 2.0 * N * N * M / (kin_f32_ms * 1e-3) / 1e12,
                        "f32_frac_of_peak": None if kin_f32_ms is None else
                        2.0 * N * N * M / (kin_f32_ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS},
            "adaptive_scan": scan_stats, "all_planes_reference": all_planes,
            "eigh_ms": eigh_ms, "model_setup_s": model_s, "setup_s": t_setup,
            "delta": float(est["delta"]), "min_p": float(np.nanmin(ps)), "device": info,
        }
        if not args.no_cpu_baseline and world == 1:      # reported on rank 0 at N=1 only
            sample = min(M, args.cpu_sample or 40 * N)
            res["cpu_baseline"] = cpu_baseline(N, sample, lmm, est, prep, ps[:sample])
    if coll is not None:
        coll.barrier()
        if rank == 0:
            try:
                os.remove(boot.path)
            except OSError:
                pass
        coll.close()
    if rank == 0:
        sys.stdout.flush()
        print(json.dumps(res))          # the ONE JSON line, last on stdout
        sys.stdout.flush()


def _device_rows(ctx, rows, n, seed):
    """Genotype rows of arbitrary global SNP indices, regenerated on the device by the counter hash."""
    tmp = ctx.geno(M=1, N=n)
    out = []
    for r in rows:
        tmp.fill_hash(seed, m_global0=int(r))
        out.append(tmp.download()[0])
    tmp.close()
    return np.vstack(out)


def cpu_baseline(N, sample, lmm, est, prep, gpu_ps):
    """The reference's loop structure (chunks of N SNPs, float32 `chunk @ M` GEMM, one
    scipy.linalg.lstsq per SNP, scipy.stats.f.sf -- linear_models.py:1315-1349) as restated in
    oracle/emmax_oracle.py:scan_loop, timed on this box's host cores on the first `sample` SNPs
    of the same workload.  Reported, not a target."""
    from scipy import linalg
    from oracle import emmax_oracle as orc
    snps = orc.hash_genotypes(0, sample, N, 20240)
    H = np.asarray(est["H_sqrt_inv"])
    h0_X = H @ lmm.X
    Q, _ = linalg.qr(h0_X, mode="economic")
    Mp = (H - Q @ (Q.T @ H)).T                                   # H'(I - QQ'), O(N^2 q)
    p = {"n": N, "q": lmm.X.shape[1], "Mp": Mp, "r": prep["r"], "h0_rss": prep["h0_rss"], "h0_betas": prep["h0_betas"]}
    t0 = time.time()
    out = orc.scan_loop(snps, p, dtype=np.float32)
    dt = time.time() - t0
    agree = float(np.nanmax(np.abs(out["ps"][:len(gpu_ps)] / gpu_ps[:len(out["ps"])] - 1)))
    blas, blas4 = "unknown", None
    try:
        import threadpoolctl
        blas = ";".join("%s:%s" % (d.get("internal_api"), d.get("num_threads")) for d in threadpoolctl.threadpool_info())
        # the reference pins its BLAS to 4 threads (linear_models.py:37): the same loop on a quarter of the sample
        with threadpoolctl.threadpool_limits(limits=4):
            t0 = time.time()
            orc.scan_loop(snps[:max(N, sample // 4)], p, dtype=np.float32)
            blas4 = max(N, sample // 4) / (time.time() - t0)
    except Exception:
        pass
    return {"value": sample / dt, "value_blas_4_threads": blas4, "unit": "SNPs/s", "cores": os.cpu_count(), "kind": "port",
            "sample": "first %d SNPs of the same workload (chunks of N), fp32 reference loop, %.1f s; BLAS %s"
                      % (sample, dt, blas),
            "max_rel_p_diff_vs_gpu": agree}


if __name__ == "__main__":
    main()

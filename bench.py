#!/usr/bin/env python
"""bench.py -- SNPs/sec of the EMMAX scan (BASELINE.json metric) on MI355X.

  python bench.py --gpus 1 --steps K --warmup W [--mode weak|strong|perm|multi]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over this rank's batch of synthetic SNPs that are already resident in HBM.
Modes (default = the BASELINE metric, unchanged since round 1):
  weak    the EMMAX scan (linear_models.py:1316-1349: per-SNP quadratic form, F statistic, p-value) of M SNPs PER
          GPU, ending with rss/F/p of every SNP on the host of every rank (N>1: after the RCCL all-gather).
          Workload at N=1: BASELINE configs[2], N=5000 x M=1,000,000 Bernoulli(0.5) genotypes (simulations.py:21-23
          restated with a counter hash, generated on the device).  The kinship is built from all G*M SNPs (partial
          counts all-reduced in HBM over RCCL), eigendecomposition + REML are replicated.
  strong  the same scan with a FIXED total of M SNPs split over the ranks (shard_range), scaling "strong".
  perm    BASELINE configs[3]: the EMMAX permutation test (linear_models.py:1125-1175) with P = 1000 permutations over
          M SNPs in total, SNP blocks sharded over the ranks; a step = t.t quadratic forms + permutation GEMM + RCCL
          MAX all-reduce of the P statistics in HBM + min_rss on every host.  value = SNP x permutation tests / s.
  multi   P phenotypes over the eigen-rotated store: a step = the HBM-bound passes of all P
          phenotypes over this rank's M SNPs (weak scaling) incl. the download of the P x M p-values; the rotation
          GEMM (once per genotype block) is reported beside it.  value = SNP x phenotype scans / s.

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
# What a bare v_mfma_i32_32x32x32_i8 loop sustains under this chip's power cap on the operands the scan GEMM feeds it
# (non-negative 7-bit digits 0..127 as A, 0/1 genotypes as B): tools/probe/mfma_digit_range.hip, DESIGN.md 4.1
# (signed base-256 digits: 4.30).  Reported beside the nominal peak, never instead of it.
I8_MFMA_SUSTAINED_TOPS = 4560.0
I8_MFMA_SUSTAINED_SOURCE = "tools/probe/mfma_digit_range.hip (digits 0..127 x 0/1 bytes, registers only, power-capped clock)"
F4_MFMA_PEAK_TOPS = 10000.0    # dense FP4 / FP6 on v_mfma_scale_f32_32x32x64_f8f6f4 (same guide: ~10 PF)
F32_MFMA_PEAK_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32 (same guide)
HBM_PEAK_GBPS = 8000.0
PCIE_GBPS = 64.0               # host link of the box (gen5 x16), the roof of anything that starts in host memory


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 3 (c5 mode: 1)")
    ap.add_argument("--warmup", type=int, default=None, help="default 1 (c5 mode: 0)")
    ap.add_argument("--mode", choices=("weak", "strong", "perm", "multi", "c5"), default="weak",
                    help="weak (default, the driver's contract): --m SNPs PER GPU; strong: --m SNPs in total split over the ranks "
                         "-- the curve to read for BASELINE config 3 (independent SNP blocks scale weakly by construction: "
                         "a healthy strong curve has the scan ~1/N and end_to_end_s flat at the replicated REML + scan model)")
    ap.add_argument("--c5-n", type=int, default=50000, help="c5 mode: individuals")
    ap.add_argument("--c5-m", type=int, default=10000000, help="c5 mode: SNPs in total, dealt chunk-wise over the ranks")
    ap.add_argument("--c5-chunk", type=int, default=50000, help="c5 mode: SNPs per chunk")
    ap.add_argument("--c5-share-of", type=int, default=0,
                    help="c5 mode on ONE GPU: time the share one rank of this many would own (c5-m / share-of SNPs, no peers)")
    ap.add_argument("--n", type=int, default=5000, help="individuals")
    ap.add_argument("--m", type=int, default=1000000, help="SNPs per GPU (weak, multi) / in total (strong, perm)")
    ap.add_argument("--perms", type=int, default=1000, help="permutations (perm mode)")
    ap.add_argument("--phenos", type=int, default=64, help="phenotypes (multi mode)")
    ap.add_argument("--digits", type=int, default=0,
                    help="digit planes of the scan model; 0 = the library default (4 planes, adaptive schedule)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-f32-kinship", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the structured-data, host-ingest and multi-phenotype sub-records of the default run")
    ap.add_argument("--cpu-sample", type=int, default=0,
                    help="SNPs in the CPU baseline sample (0 = all M SNPs of the workload at the default size, "
                         "~80 s of CPU work at N=5000, M=1M)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="launcher check: every rank prints its rendezvous environment as JSON and exits without "
                         "loading the HIP library or touching a GPU")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 1 if args.mode == "c5" else 3
    if args.warmup is None:
        args.warmup = 0 if args.mode == "c5" else 1
    return args


# ------------------------------------------------------------------------------------------------- self-launch
def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: start one child process per GPU
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT / MMG_RUN_ID, the variables torch.distributed.run
    exports and mixmogam_amd.dist.file_bootstrap keys the RCCL rendezvous on), forward rank 0's stdout -- whose last
    line is the ONE JSON line -- and return non-zero if any rank does.  This parent never loads libmixmogam_hip and
    never touches a GPU; nothing is exec'ed; children are stopped by their exact PIDs."""
    import socket
    import subprocess
    import threading
    world = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    run_id = "self%d_%d" % (os.getpid(), int(time.time()))
    procs = []
    for r in range(world):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "MMG_RUN_ID": run_id})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE, stderr=None))
    lines = [[] for _ in range(world)]

    def pump(r):
        for raw in procs[r].stdout:
            line = raw.decode(errors="replace")
            lines[r].append(line)
            if r != 0 and not args.dry_launch:        # other ranks' chatter must not follow rank 0's JSON line on stdout
                sys.stderr.write("[rank %d] %s" % (r, line))
    threads = [threading.Thread(target=pump, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    rc = 0
    alive = set(range(world))
    while alive:                                      # a rank that dies leaves the others inside an RCCL call for good
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                sys.stderr.write("bench.py: rank %d exited with code %d; stopping the other ranks\n" % (r, code))
                for o in sorted(alive):
                    procs[o].terminate()
        time.sleep(0.05)
    for t in threads:
        t.join(timeout=5)
    if args.dry_launch:
        recs = [json.loads(l) for r in range(world) for l in lines[r] if l.startswith("{")]
        print(json.dumps({"dry_launch": True, "world": world, "ranks": recs}))
        return rc
    sys.stdout.write("".join(lines[0]))
    sys.stdout.flush()
    return rc


def _check_distinct_devices(coll, rank, world, local_rank, info):
    """[{rank, device, pci_bus_id}] of every rank (all-gathered over RCCL as numbers); exits non-zero on every rank when two
    ranks report the same PCI bus id."""
    def code(bus):                                         # "0000:c1:00.0" -> an exact integer in a double
        try:
            dom, b, rest = bus.split(":")
            dev, fn = rest.split(".")
            return float((int(dom, 16) << 24) | (int(b, 16) << 16) | (int(dev, 16) << 8) | int(fn, 16))
        except Exception:
            return -1.0 - rank
    mine = np.array([float(rank), float(local_rank), code(info["pci_bus_id"])])
    rows = np.asarray(coll.allgather(mine)).reshape(world, 3) if coll is not None else mine.reshape(1, 3)
    ident = [{"rank": int(r), "device": int(d), "pci_code": int(c)} for r, d, c in rows]
    ident[rank]["pci_bus_id"] = info["pci_bus_id"]
    codes = [i["pci_code"] for i in ident]
    if len(set(codes)) != len(codes):
        raise SystemExit("bench.py: two ranks sit on the same GPU (PCI codes %s): refusing to time anything" % codes)
    return ident


def rccl_probe(ctx, coll, comm_h, N, m_block, repeats=3):
    """The two exchange steps of the sharded hot path, timed on their own before the bench (first contact with RCCL over
    xGMI): the all-reduce SUM of the N x N fp64 kinship accumulator in HBM (mmg_kin_acc_allreduce; 200 MB at C3 -- what
    mmg_kinship_ibs_i8_sharded does after its GEMM) and the all-gather of one rank's (rss, F, p) block (mmg_comm_allgather_f64,
    host-staged like mmg_comm_allgather_scan's delivery).  busbw = algbw x 2 (w - 1) / w (ring convention), 0 for one rank."""
    w = coll.world
    acc = ctx.kinship_accumulator(N)
    t = []
    for _ in range(repeats + 1):                           # the first call builds RCCL's channels
        coll.barrier()
        t0 = time.time()
        acc.allreduce(comm_h)
        coll.barrier()
        t.append(time.time() - t0)
    acc.close()
    ar = max(min(t[1:]), 1e-9)                                 # (a stand-in collective in the tests can take no measurable time)
    nbytes = 8.0 * N * N
    blk = np.zeros(3 * min(int(m_block), 1 << 22))
    tg = []
    for _ in range(repeats + 1):
        coll.barrier()
        t0 = time.time()
        coll.allgather(blk)
        coll.barrier()
        tg.append(time.time() - t0)
    ag = max(min(tg[1:]), 1e-9)
    return {"nranks": w, "allreduce_bytes": nbytes, "allreduce_ms": 1e3 * ar, "allreduce_first_call_ms": 1e3 * t[0],
            "allreduce_algbw_gbps": nbytes / ar / 1e9, "allreduce_busbw_gbps": nbytes / ar / 1e9 * 2.0 * (w - 1) / w,
            "allgather_bytes_per_rank": 8.0 * blk.size, "allgather_ms": 1e3 * ag, "allgather_first_call_ms": 1e3 * tg[0],
            "allgather_algbw_gbps": 8.0 * blk.size * w / ag / 1e9,
            "note": "N x N fp64 accumulator summed in HBM (RCCL over xGMI) and one (rss, F, p) block gathered through the "
                    "host-staged path, each with a barrier on both sides; per-link xGMI ~153 GB/s"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))               # the parent of the ranks: no GPU, no HIP library
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.dry_launch:
        print(json.dumps({k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR",
                                                         "MASTER_PORT", "MMG_RUN_ID")}))
        return
    os.environ.setdefault("MMG_DEVICE", str(local_rank))

    from mixmogam_amd import _lib, dist as mdist, kinship, linear_models as lm

    n_dev = _lib.device_count()
    if n_dev < world:
        raise SystemExit("bench.py: --gpus %d needs %d devices, this box has %d (rank %d)" % (world, world, n_dev, rank))

    # No torch in this process: libmixmogam_hip links the system ROCm runtime and importing
    # torch's bundled one beside it segfaults.  Rendezvous = rank 0's ncclUniqueId through a
    # per-user file keyed by the launcher's MASTER_PORT; barriers and the max-over-ranks time go
    # over RCCL itself.
    ctx = _lib.Context(local_rank)
    info = ctx.device_info()
    coll, rccl_nranks = None, 1
    if world > 1 or os.environ.get("MMG_BENCH_FORCE_COLL"):   # the env knob exercises the RCCL path on 1 GPU
        coll = mdist.RcclCollectives(ctx, rank, world, mdist.file_bootstrap(rank, world))
        rccl_nranks = coll.info()[2]
        if rccl_nranks != world:
            raise SystemExit("RCCL reports %d ranks, launcher %d" % (rccl_nranks, world))
    comm_h = coll.device_comm if coll is not None else None
    # first contact with a multi-GPU node: say where this rank sits BEFORE anything is timed, and refuse to go on when two ranks
    # share a GPU (LOCAL_RANK not honoured, a visibility mask ...): a scaling curve measured that way is worse than none
    sys.stderr.write("[bench rank %d/%d] device %d  pci %s  %s  %d CUs  %.0f GB\n"
                     % (rank, world, local_rank, info["pci_bus_id"], info["arch"], info["n_cu"], info["hbm_bytes"] / 2.0 ** 30))
    sys.stderr.flush()
    identity = _check_distinct_devices(coll, rank, world, local_rank, info)
    rccl = rccl_probe(ctx, coll, comm_h, args.n if args.mode != "c5" else args.c5_n,
                      args.m if args.mode != "c5" else args.c5_chunk) if coll is not None else None

    if args.mode == "c5":
        return bench_c5(args, ctx, coll, rank, world, info, rccl_nranks, identity, rccl)
    N, D, mode = args.n, args.digits, args.mode
    if mode in ("weak", "multi"):
        M, Mtot, m_global0 = args.m, args.m * world, rank * args.m
    else:
        m0, m1 = mdist.shard_range(args.m, rank, world)
        M, Mtot, m_global0 = m1 - m0, args.m, m0
    t_setup = time.time()

    # ---- synthetic genotypes, generated in HBM (this rank's block of the global SNP axis)
    g = ctx.geno(M=M, N=N)
    g.fill_hash(20240, m_global0=m_global0, thr16=32768)

    # ---- phenotype (simulations.py:64-85 restated): 100 causal SNPs of the global data set
    rng = np.random.RandomState(20241)
    causal = np.sort(rng.choice(Mtot, 100, replace=False))
    effects = rng.exponential(1.0, size=100)
    from_hash = _device_rows(ctx, causal, N, 20240)
    gen = effects @ from_hash.astype(np.float64)
    err = rng.normal(0, 1, size=N)
    y = gen + err * np.sqrt((0.2 / 0.8) * (np.var(gen, ddof=1) / np.var(err, ddof=1)))
    y = (y - y.mean()) / y.std()

    # ---- kinship: exact IBS counts on the int8 matrix cores, partial counts of the ranks summed in HBM over RCCL
    # (+ the fp32-MFMA twin of the north star on this rank's block, for the TFLOP/s figure)
    def pack_ms():                                          # 0 when the call needed no image pass (round 3: the IBS
        try:                                                # GEMM reads the SNP-major store through transposed LDS reads)
            return ctx.kernel_ms("pack")
        except _lib.MixmogamHipError:
            return 0.0
    fp4_kin = os.environ.get("MMG_KIN_FP4", "1") != "0" and not os.environ.get("MMG_KIN_KERNEL")
    counts = ctx.kinship_ibs_counts(g, comm=comm_h)
    kin_i8_first_ms = ctx.kernel_ms("kinship")              # first launch of the kernel: includes its code-object load
    counts2 = ctx.kinship_ibs_counts(g, comm=comm_h)
    if not np.array_equal(counts, counts2):
        raise SystemExit("kinship counts differ between two calls")
    del counts2
    kin_i8_ms, kin_i8_pack_ms = ctx.kernel_ms("kinship"), pack_ms()
    if fp4_kin:                                             # the FP4 call times image pass + GEMM together
        kin_i8_ms -= kin_i8_pack_ms
    kin_f32_ms = kin_f32_pack_ms = None
    if not args.no_f32_kinship:
        cf = ctx.kinship_affine(g)
        kin_f32_ms, kin_f32_pack_ms = ctx.kernel_ms("kinship"), pack_ms()
        if world == 1 and not np.array_equal(cf, counts.astype(np.float64)):
            raise SystemExit("fp32-MFMA and int8-MFMA kinship counts differ")
        del cf
    grm = None
    if not args.no_f32_kinship and world == 1 and mode == "weak":
        # the GRM through the exact int8 route (the chunked drivers' kinship): per-SNP weights 1/std^2 as int8 digits
        acc = ctx.kinship_accumulator(N)
        wall = []
        for _ in range(3):                                   # the first call allocates the accumulator's workspace
            t0 = time.time()
            acc.add_grm(g)
            wall.append(1e3 * (time.time() - t0))
        grm = {"wall_ms": min(wall[1:]), "first_call_wall_ms": wall[0], "digit_plane_gemms_ms": ctx.kernel_ms("grm"),
               "pack_ms": pack_ms(),
               # executed: 4 planes x the 128-tiles of the upper 256-tile triangle x 2 ops per MAC
               "executed_int8_tops": (4.0 * 2.0 * 256.0 * 256.0 * ((-(-N // 256)) * (-(-N // 256) + 1) / 2) * M
                                      / (ctx.kernel_ms("grm") * 1e-3) / 1e12),
               "kernel": "kinship_grm4_kernel" if os.environ.get("MMG_GRM_FUSED", "1") != "0" else "kinship_i8_tr_kernel x 4",
               "note": "mmg_kin_acc_add_grm: z z' = a^2 s s' + ab(s 1' + 1 s') + b^2 1 1', the weighted Gram matrix as 4 "
                       "exact int8 digit planes of the weights -- all four from ONE pass over the plain genotype tiles "
                       "(kinship_grm4_kernel: the digit scaling is a byte mask in registers; pack_ms = the weighted column "
                       "sums); MMG_GRM_FUSED=0: one transposed-read GEMM per plane over digit-scaled images; compare "
                       "kinship_f32_kernel"}
        acc.close()
    K = kinship.scale_k(counts.astype(np.float64) / (2.0 * Mtot) + 0.5)

    # ---- eigh + REML (replicated), model -> device.  --mode perm (round 5): no eigendecomposition anywhere -- REML through the
    # band reduction, the scan model and the permutation plan from the Cholesky factor in HBM (bench_perm)
    lmm = lm.LinearMixedModel(y, ctx=ctx)
    lmm.add_random_effect(K)
    t0 = time.time()
    eig_L = est = prep = None
    eigh_ms = None
    if mode != "perm":
        eig_L = lmm._get_eigen_L_()
        eigh_ms = ctx.kernel_ms("eigh")
        est = lmm.get_estimates(eig_L, method="REML")       # REML sums from eig_L alone: no second eigh
        prep = lmm.scan_prepare(est["H_sqrt_inv"])
        ctx.scan_set_model(prep["A"], prep["w"], D)
        n_p = prep["n_p"]
    model_s = time.time() - t0
    t_setup = time.time() - t_setup

    def barrier():
        # every library call above blocks on its HIP stream (hipStreamSynchronize), so the device
        # is idle here; across ranks: an RCCL all-reduce.
        if coll is not None:
            coll.barrier()

    common = {"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True,
              "vs_baseline": None, "data": "synthetic", "rccl_nranks": rccl_nranks, "mode": mode,
              "devices": identity, "rccl": rccl,
              "rccl_allreduce_gbps": rccl["allreduce_busbw_gbps"] if rccl else None}

    if mode == "perm":
        res = bench_perm(args, ctx, coll, comm_h, g, lmm, y, N, M, Mtot, barrier, common)
    elif mode == "multi":
        res = bench_multi(args, ctx, coll, g, lmm, eig_L, N, M, Mtot, barrier, common)
    else:
        res = bench_scan(args, ctx, coll, comm_h, g, lmm, est, prep, N, M, Mtot, D, barrier, common, rank, world,
                         kin_i8_ms, kin_i8_pack_ms, kin_f32_ms, kin_f32_pack_ms, fp4_kin, kin_i8_first_ms)
        if res is not None and grm is not None:
            res["roofline_kinship"]["grm_exact_i8"] = grm
    # What the timed region does not show: kinship is sharded, but eigh + REML + the scan model are
    # REPLICATED on every rank, and a caller of linear_models.emmax() pays for them.  end_to_end_emmax_s = one
    # lm.emmax()-equivalent call (LinearMixedModel -> emmax_f_test: eigh, REML, device scan model, scan of this rank's
    # resident SNPs, results on the host) per rank; both are gathered so a scaling curve can be read against them.
    e2e = e2e_eig = None
    if mode in ("weak", "strong"):
        e2e_first = None
        for _ in range(3):                                 # the first call pays library start-up (reported as
            t0 = time.time()                               # ..._first_call_s); best of the two after it
            lmm2 = lm.LinearMixedModel(y, ctx=ctx)
            lmm2.add_random_effect(K)
            r2 = lmm2.emmax_f_test(g, emma_num=0)
            dt = time.time() - t0
            if e2e_first is None:
                e2e_first = dt
            else:
                e2e = dt if e2e is None else min(e2e, dt)
            e2e_timings = r2.get("timings")
            del r2, lmm2
        # the O(N^3) stage of that route on its own: band reduction of K (csrc/reml_band.hip + dense64.hip), 4 N^3 / 3 flop
        band_rec = None
        try:
            rw = ctx.reml(K, np.ones((N, 1)), y)
            runs = []
            for _ in range(6):                                 # (the first reduction of a process loads code; best of the other five:
                rw.close()                                     # a single reduction varies by +-0.5 ms with the clock the box holds)
                rw = ctx.reml(K, np.ones((N, 1)), y)
                rw.sums(np.array([1.0]), route="band")
                runs.append(rw.band_info())
            bi = min(runs[1:], key=lambda v: v["seconds"])
            bi = dict(bi, seconds_median=sorted(v["seconds"] for v in runs[1:])[2])
            rw.close()
            flop = 4.0 * N ** 3 / 3.0
            band_rec = {"kernel": "band_reduce_cqr (gram_slices / cholqr_head1|2 / rows_gemm / sym_skinny / nt_update_lower kernels)",
                        "n": N, "seconds": bi["seconds"], "seconds_median_of_5": bi["seconds_median"],
                        "householder_fallback": bi["householder_fallback"],
                        "algorithmic_flop": flop, "achieved": flop / bi["seconds"] / 1e12, "peak": 78.6, "unit": "TFLOP/s",
                        "frac": flop / bi["seconds"] / 1e12 / 78.6, "bound": "latency at this N (77 panels x 13 dependent launches whose work is "
                        "microseconds; 30-33 TFLOP/s at N = 50,000, DESIGN 4.4)"}
        except Exception as e:                                 # never let an extra break the line
            band_rec = {"error": str(e)}
        # the same call on the eigendecomposition route (rocSOLVER dsyevd + REML from eig_L): what round 3's default was
        keep = lm.EIGEN_FREE_MIN_N
        lm.EIGEN_FREE_MIN_N = 1 << 30
        try:
            for _ in range(2):
                t0 = time.time()
                lmm2 = lm.LinearMixedModel(y, ctx=ctx)
                lmm2.add_random_effect(K)
                r2 = lmm2.emmax_f_test(g, emma_num=0)
                dt = time.time() - t0
                e2e_eig = dt if e2e_eig is None else min(e2e_eig, dt)
                e2e_eig_timings = r2.get("timings")
                del r2, lmm2
        finally:
            lm.EIGEN_FREE_MIN_N = keep
    # ---- the whole job as one clock (the timed region above leaves the replicated stages out): IBS kinship of
    # every rank's SNP block (RCCL sum of the counts, scale_k on the device, kept in HBM) -> REML -> scan model -> scan of this
    # rank's SNPs -> p-values on the host; max over the ranks
    job_s = job_phases = None
    if mode in ("weak", "strong"):
        def whole_job():
            t0 = time.time()
            K2 = kinship.calc_ibs_kinship(None, geno=g, ctx=ctx, keep_device=True, comm=comm_h, m_total=Mtot)
            tk = time.time()
            lmm2 = lm.LinearMixedModel(y, ctx=ctx)
            lmm2.add_random_effect(K2)
            r2 = lmm2.emmax_f_test(g, emma_num=0)
            K2.close()
            t1 = time.time()
            ph = {"kinship": tk - t0}
            ph.update({k: v for k, v in r2.get("timings", {}).items() if v})
            return t1 - t0, ph
        whole_job()
        barrier()
        runs = [whole_job() for _ in range(2)]
        barrier()
        job_s, job_phases = min(runs, key=lambda v: v[0])
        if coll is not None:
            job_s = float(coll.allreduce(np.array([job_s]), "max")[0])
    setup_per_rank = [t_setup]
    e2e_per_rank = [e2e]
    if coll is not None:
        setup_per_rank = [float(v) for v in coll.allgather(np.array([t_setup]))]
        if e2e is not None:
            e2e_per_rank = [float(v) for v in coll.allgather(np.array([e2e]))]
    if rank == 0:
        res.update({"model_setup_s": model_s, "setup_s": t_setup, "device": info, "setup_s_per_rank": setup_per_rank,
                    "setup_note": "genotype fill + kinship (SNP-sharded, all-reduced) + eigh + REML + scan model "
                                  "(replicated on every rank), outside the timed region" if mode != "perm" else
                                  "genotype fill + kinship (SNP-sharded, all-reduced), outside the timed region; REML, the scan "
                                  "model and the permutation plan are in model_and_plan_s (no eigendecomposition)"})
        if est is not None:
            res.update({"eigh_ms": eigh_ms, "delta": float(est["delta"])})
        if job_s is not None:
            res.update({"end_to_end_s": job_s, "end_to_end_phases_s": job_phases, "end_to_end_snps_per_s": Mtot / job_s,
                        "end_to_end_s_note": "kinship (IBS counts of every rank's SNP block, RCCL sum, scale_k on the device, "
                                           "kept in HBM) -> REML (band reduction, replicated) -> scan model (replicated) -> "
                                           "EMMAX scan of this rank's SNPs -> p-values on the host; best of two warm runs, "
                                           "max over the ranks.  This is the clock a scaling curve should be read against: "
                                           "`value` times the scan alone"})
        if e2e is not None:
            res.update({"end_to_end_emmax_s": max(e2e_per_rank), "end_to_end_emmax_s_per_rank": e2e_per_rank,
                        "end_to_end_emmax_first_call_s": e2e_first, "end_to_end_emmax_phases_s": e2e_timings,
                        "end_to_end_emmax_route": "eigendecomposition-free: band reduction of K on own kernels, REML "
                                                  "search on an interpolant of the sums, own blocked "
                                                  "Cholesky for the scan model" if e2e_timings and e2e_timings.get("eig_L") == 0.0
                                                  else "eigh (rocSOLVER dsyevd)",
                        "reml_band_reduction": band_rec,
                        "end_to_end_emmax_eigen_route_s": e2e_eig,
                        "end_to_end_emmax_eigen_route_phases_s": e2e_eig_timings,
                        "end_to_end_note": "one lm.emmax()-equivalent call on resident genotypes (LinearMixedModel -> "
                                           "emmax_f_test -> results on the host), best of two warm calls; the eigen "
                                           "route is the same call with linear_models.EIGEN_FREE_MIN_N raised"})
    if coll is not None:
        coll.barrier()
        coll.close()
    if rank == 0:
        sys.stdout.flush()
        print(json.dumps(res))          # the ONE JSON line, last on stdout
        sys.stdout.flush()


# ----------------------------------------------------------------------------------------------- the BASELINE metric
def bench_scan(args, ctx, coll, comm_h, g, lmm, est, prep, N, M, Mtot, D, barrier, common, rank, world, kin_i8_ms,
               kin_i8_pack_ms, kin_f32_ms, kin_f32_pack_ms, fp4_kin=True, kin_i8_first_ms=None):
    from mixmogam_amd import dist as mdist
    n_p = prep["n_p"]
    # Result buffers are allocated once and page-locked (mmg_host_alloc).  Delivery is double buffered: the
    # (rss, F, p) of step i are snapshotted on the device and all-gathered over RCCL / downloaded on a second
    # HIP stream (mmg_scan_deliver_begin) while step i+1 scans; the last delivery is awaited inside the timed
    # region, so every step's results are on the host of every rank before the closing barrier.
    count = M if args.mode == "weak" else max(b - a for a, b in (mdist.shard_range(args.m, r, world) for r in range(world)))
    outs2 = [[ctx.pinned_empty(count * world) for _ in range(3)] for _ in range(2)]

    def step(i):
        ctx.scan(g, prep["h0_rss"], n_p, fetch=False)           # blocks until the kernels finish
        ctx.scan_deliver_begin(outs2[i & 1], count=count, comm=comm_h)
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
    per_rank_quad = [float(np.mean(quad_ms))]
    if coll is not None:
        elapsed = float(coll.allreduce(np.array([elapsed]), "max")[0])
        per_rank_quad = [float(v) for v in coll.allgather(np.array([np.mean(quad_ms)]))]

    ps = out[2][rank * count:rank * count + M].copy()
    scan_stats = ctx.scan_last_stats()
    # For the record, outside the timed region: the same scan with all four digit planes for every SNP (what the
    # adaptive schedule is measured against) -- its time and how far the two sets of p-values are apart.
    all_planes = None
    if scan_stats["adaptive"] and rank == 0:
        all_planes = _all_planes_reference(ctx, g, prep, n_p, ps)
        ctx.scan_set_model(prep["A"], prep["w"], D)
    if rank != 0:
        return None
    Npad = -(-N // 256) * 256
    nJ = Npad // 256
    Dn = D if D else 4
    plane_ops = 2.0 * 256.0 * 256.0 * 256.0 * (nJ * (nJ + 1) / 2)          # one digit plane over one 256-SNP block

    def exec_ops_of(stats, m):
        if stats["adaptive"] and not stats["fell_back"]:                   # 3 planes for all + 1 for the refined
            return plane_ops * ((Dn - 1) * (-(-m // 256)) + (-(-stats["n_refined"] // 256)))
        return plane_ops * Dn * (-(-m // 256)) * (2 if stats["fell_back"] else 1)

    def roofline_of(qms, m, stats):
        alg_flop = (2.0 * N * N + 4.0 * N) * m                 # SURVEY 8d per-SNP figure x SNPs per launch
        achieved = alg_flop / (qms * 1e-3) / 1e12
        ex = exec_ops_of(stats, m)
        return {"bound": "mfma", "kernel": QUAD_KERNEL, "achieved": achieved, "peak": I8_MFMA_PEAK_TOPS,
                "unit": "TFLOP/s", "frac": achieved / I8_MFMA_PEAK_TOPS, "ms": qms,
                "algorithmic_bytes": float(-(-m // 256) * 256 * Npad + Dn * Npad * Npad),
                "executed_int8_tops": ex / (qms * 1e-3) / 1e12,
                "executed_frac": ex / (qms * 1e-3) / 1e12 / I8_MFMA_PEAK_TOPS,
                "executed_frac_of_sustained_mfma_rate": ex / (qms * 1e-3) / 1e12 / I8_MFMA_SUSTAINED_TOPS,
                "sustained_mfma_rate_tops": I8_MFMA_SUSTAINED_TOPS, "sustained_mfma_rate_source": I8_MFMA_SUSTAINED_SOURCE}

    traffic = _profiled_traffic(N, M, D, scan_stats["adaptive"])
    roof = roofline_of(per_rank_quad[0], M, scan_stats)
    tq = traffic.get(QUAD_KERNEL)
    # what north_star asks to see beside the MFMA fraction: HBM-side GB/s of the dominant kernel (counter bytes of the committed
    # --pmc passes over THIS run's hipEvent time) and how much of that traffic the algorithm did not ask for
    roof.update({"hbm_gbps_achieved": (tq / (roof["ms"] * 1e-3) / 1e9) if tq else None,
                 "hbm_frac_of_peak": (tq / (roof["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBPS) if tq else None,
                 "wasted_traffic_ratio": (tq / roof["algorithmic_bytes"]) if tq else None,
                 "ms_rocprof_first_pass_mean": traffic.get(QUAD_KERNEL + ":first_pass_ms_mean"),
                 "ms_rocprof_source": "profiles/r6_bench_c3_scan_launches.txt (per-launch --kernel-trace durations of this "
                                      "command; first passes = the timed kernel)"})
    roof.update({"traffic": tq, "traffic_from_profiles": True,
                 "traffic_unit": "HBM-side bytes per launch from committed rocprofv3 --pmc passes of this command "
                                 "(profiles/traffic_c3.json: FETCH_SIZE, WRITE_SIZE in separate passes, gfx950 "
                                 "corrections) -- NOT measured in this run; null when the config differs"})
    ms_per_step = 1e3 * elapsed / args.steps
    kin_exec = 2.0 * 256.0 * 256.0 * (nJ * (nJ + 1) / 2) * M     # lower-triangle tiles x contraction length
    kin_sym = float(N) * (N + 1) * M                             # the symmetric product: N(N+1)/2 entries x 2M flop
    kin_full = 2.0 * N * N * M                                   # what the reference forms (kinship.py:44)

    def kin_roof(name, gemm_ms, peak, unit, pass_ms=0.0, pass_name=None):
        if gemm_ms is None:
            return None
        ms = gemm_ms + (pass_ms or 0.0)                          # everything a caller pays for on the device
        d = {"bound": "mfma", "kernel": name, "ms": ms, "ms_gemm": gemm_ms, "achieved": kin_sym / (ms * 1e-3) / 1e12,
             "peak": peak, "unit": unit, "frac": kin_sym / (ms * 1e-3) / 1e12 / peak,
             "executed": kin_exec / (ms * 1e-3) / 1e12, "executed_frac": kin_exec / (ms * 1e-3) / 1e12 / peak,
             "executed_frac_gemm_only": kin_exec / (gemm_ms * 1e-3) / 1e12 / peak,
             "vs_reference_full_product": kin_full / (ms * 1e-3) / 1e12,
             "algorithmic_flop": kin_sym, "executed_flop": kin_exec, "reference_full_product_flop": kin_full,
             "algorithmic_bytes": float(M * Npad + 8.0 * N * N), "traffic": traffic.get(name),
             "traffic_from_profiles": True,
             "note": "algorithmic = the symmetric product N(N+1)M (K = K'); the reference forms all 2 N^2 M "
                     "(kinship.py:44): `vs_reference_full_product` is that figure over the same time, a speed-up "
                     "statement, not a roofline fraction; executed = the lower triangle of 256^2 tiles actually run; "
                     "`ms` includes every device pass the call needs (ms_gemm + the image pass: the individual-major "
                     "copy of the fp32 kernel; the exact-count kernel reads the store's own E2M1 twin, written by the "
                     "ingest / generator kernels -- round 4 -- so its call has no image pass unless MMG_FP4_TWIN=0)"}
        if pass_name is not None:
            d[pass_name] = pass_ms
        return d

    res = dict(common)
    res.update({
        "metric": "SNPs/sec EMMAX scan", "value": Mtot * args.steps / elapsed, "unit": "SNPs/s",
        "ms_per_step": ms_per_step, "scaling": "weak" if args.mode == "weak" else "strong", "dtype": "i8",
        "config": {"workload": "EMMAX scan N=%d individuals x M=%d SNPs %s (BASELINE configs[2] shape), "
                               "Bernoulli(0.5) hash genotypes resident in HBM, q=1"
                               % (N, args.m, "per GPU" if args.mode == "weak" else "in total, split over the ranks"),
                   "n_individuals": N, "snps_per_gpu": M, "snps_total": Mtot, "digits": Dn,
                   "digit_schedule": "adaptive (3 planes for every SNP, the 4th where p could move by 2.5e-7)"
                                     if scan_stats["adaptive"] else "all planes for every SNP",
                   "parallelism": "snp-block x%d" % world,
                   "delivery": "double buffered: step i's results are gathered (RCCL) / downloaded on a second "
                               "stream while step i+1 scans; the last one is awaited inside the timed region",
                   },
        "roofline": roof,
        "roofline_per_rank": [roofline_of(q, M, scan_stats)["frac"] for q in per_rank_quad],
        "scan_quad_ms_per_rank": per_rank_quad,
        "finalize_kernel": _finalize_record(N, Npad, M, float(np.mean(fin_ms))),
        # second headline metric: kinship GEMM TFLOP/s vs MFMA peak, one record per kernel
        "roofline_kinship": {"f32": kin_roof("kinship_f32_kernel", kin_f32_ms, F32_MFMA_PEAK_TFLOPS, "TFLOP/s",
                                             kin_f32_pack_ms, "transpose_pass_ms"),
                             # binary stores run the exact counts on FP4 operands (0/1 are exact in E2M1, fp32
                             # accumulators hold exact integers): priced against the FP4 peak, not the int8 one
                             "i8": dict(kin_roof("kinship_f4_tr_kernel", kin_i8_ms, F4_MFMA_PEAK_TOPS, "TOP/s",
                                                 kin_i8_pack_ms, "fp4_image_pass_ms")
                                        if fp4_kin else
                                        kin_roof("kinship_i8_tr_kernel", kin_i8_ms, I8_MFMA_PEAK_TOPS, "TOP/s",
                                                 kin_i8_pack_ms, "transpose_pass_ms"),
                                        first_call_ms=kin_i8_first_ms,
                                        fp4_twin_store=os.environ.get("MMG_FP4_TWIN", "1") != "0")},
        "adaptive_scan": scan_stats, "all_planes_reference": all_planes, "min_p": float(np.nanmin(ps)),
    })
    if world == 1 and not args.no_extras:
        res["structured"] = structured_record(ctx, N, M, exec_ops_of)
        res["host_ingest"] = ingest_record(ctx, g, prep, n_p, N, M, ms_per_step)
        res["multi_phenotype"] = multi_record(ctx, g, lmm, N, M)
    if not args.no_cpu_baseline and world == 1:      # reported on rank 0 at N=1 only
        sample = min(M, args.cpu_sample or M)            # default: every SNP of the workload (the full C3)
        res["cpu_baseline"] = cpu_baseline(N, sample, lmm, est, prep, ps[:sample], g)
    return res


def _finalize_record(N, Npad, M, ms):
    """The per-SNP pass after the quadratic-form GEMM.  On binary stores with >= 16 padding rows (1 <= N mod 256 <=
    240; api.hip:scan_lin_usable -- true for the hash / structured genotypes of this bench) the linear terms come out
    of the GEMM and scan_finalize_lin_kernel moves 16 int32 + 8 B q in and ~5 doubles out per SNP; otherwise
    scan_finalize_kernel re-reads the store (Npad + 56 B per SNP)."""
    lin = 1 <= N % 256 <= 240
    b = (16 * 4 + 8 + 6 * 8.0) if lin else (Npad + 56.0)
    return {"ms": ms, "kernel": "scan_finalize_lin_kernel" if lin else "scan_finalize_kernel", "bytes_per_snp": b,
            "hbm_gbps": M * b / (ms * 1e-3) / 1e9, "peak_gbps": HBM_PEAK_GBPS,
            "note": "launch-latency sized on the lin path (0.05 ms): not a bandwidth statement"}


def _all_planes_reference(ctx, g, prep, n_p, ps_adaptive):
    ctx.scan_set_model(prep["A"], prep["w"], 4)
    ctx.scan(g, prep["h0_rss"], n_p, fetch=False)
    t1 = time.time()
    for _ in range(2):
        ctx.scan(g, prep["h0_rss"], n_p, fetch=False)
    dt_all = (time.time() - t1) / 2
    ref_out = ctx.scan(g, prep["h0_rss"], n_p)
    ok = ref_out["ps"] > 1e-290
    return {"repeats": 2, "ms_per_scan_kernels_only": 1e3 * dt_all, "scan_quad_ms": ctx.kernel_ms("scan_quad"),
            "snps_per_s_kernels_only": len(ps_adaptive) / dt_all,
            "max_rel_p_diff_adaptive_vs_all_planes": float(np.max(np.abs(ps_adaptive[ok] / ref_out["ps"][ok] - 1)))}


def _profiled_traffic(N, M, D, adaptive):
    """HBM bytes per launch measured with rocprofv3 PMC passes of this same command (profiles/traffic_c3.json)."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_c3.json")))
        # `--digits 0` (the default model) runs four digit planes with the adaptive schedule: the profile's "digits": 4
        # (rounds 2 and 3 both compared the dictionaries literally and shipped `traffic: null`; tests/test_host.py now
        # loads the committed file with the default arguments)
        cfg = tj["config"]
        if (cfg["n"], cfg["m"], cfg.get("digits", 4)) == (N, M, D or 4) and tj.get("adaptive", False) == adaptive:
            out = {k: v.get("hbm_bytes_corrected") for k, v in tj["kernels"].items()}
            for k, v in tj["kernels"].items():                 # tools/scan_launch_list.py: mean --kernel-trace duration of the first passes
                if v.get("first_pass_ms_mean") is not None:
                    out[k + ":first_pass_ms_mean"] = v["first_pass_ms_mean"]
            return out
    except Exception:
        pass
    return {}


def structured_record(ctx, N, M, exec_ops_of):
    """The same scan on STRUCTURED genotypes (3 populations, Fst ~ 0.04: interior REML optimum, correlated SNPs) with
    120 strong causal SNPs -- the regime real GWAS lives in, where the adaptive digit schedule has to refine far more
    SNPs than on the null-like Bernoulli data of the headline: refined count, time, and the distance to the
    all-planes scan; `value_all_planes` is what a user sees if the refinement stops paying."""
    from mixmogam_amd import kinship, linear_models as lm
    g = ctx.geno(M=M, N=N)
    g.fill_structured(20250, 0, npop=3, spread_q16=9830)
    counts = ctx.kinship_ibs_counts(g)
    K = kinship.scale_k(counts.astype(np.float64) / (2.0 * M) + 0.5)
    rng = np.random.RandomState(20251)
    causal = np.sort(rng.choice(M, 120, replace=False))
    rows = g.download_rows(causal).astype(np.float64)
    gen = rng.exponential(1.0, size=120) @ (rows - rows.mean(1, keepdims=True))
    err = rng.normal(0, 1, size=N)
    y = gen + err * np.sqrt((0.2 / 0.8) * (np.var(gen, ddof=1) / np.var(err, ddof=1)))
    y = (y - y.mean()) / y.std()
    lmm = lm.LinearMixedModel(y, ctx=ctx)
    lmm.add_random_effect(K)
    est = lmm.get_estimates(lmm._get_eigen_L_(), method="REML")
    prep = lmm.scan_prepare(est["H_sqrt_inv"])
    ctx.scan_set_model(prep["A"], prep["w"], 0)
    ctx.scan(g, prep["h0_rss"], prep["n_p"], fetch=False)
    t0 = time.time()
    reps = 3
    for _ in range(reps):
        ctx.scan(g, prep["h0_rss"], prep["n_p"], fetch=False)
    dt = (time.time() - t0) / reps
    stats = ctx.scan_last_stats()
    qms = ctx.kernel_ms("scan_quad")
    ada = ctx.scan(g, prep["h0_rss"], prep["n_p"])
    allp = _all_planes_reference(ctx, g, prep, prep["n_p"], ada["ps"])
    g.close()
    return {"workload": "N=%d x M=%d, 3 populations (mmg_geno_fill_structured, spread 0.15), 120 causal SNPs, h2 0.8" % (N, M),
            "pseudo_heritability": float(est["pseudo_heritability"]), "min_p": float(ada["ps"].min()),
            "n_p_below_1e-8": int((ada["ps"] < 1e-8).sum()), "adaptive_scan": stats,
            "repeats": reps, "ms_per_scan_kernels_only": 1e3 * dt, "scan_quad_ms": qms, "value_adaptive": M / dt,
            "value_all_planes": allp["snps_per_s_kernels_only"], "all_planes_reference": allp,
            "executed_frac": exec_ops_of(stats, M) / (qms * 1e-3) / 1e12 / I8_MFMA_PEAK_TOPS}


def ingest_record(ctx, g, prep, n_p, N, M, ms_per_step):
    """the scan number WITH the host->device transfer of the genotypes, reported separately -- for int8
    genotypes (5 GB at C3) and for the config's fp32 [M x N] genotypes (20 GB, converted to the int8 store on the
    device).  `value_h2d_inclusive*` is never `value`."""
    from mixmogam_amd import hdf5_data
    # the headline model with its adaptive schedule (structured_record leaves ITS model, all planes, in the context --
    # rounds 2-3 measured the pipelined figures below with that leftover: 41 instead of 33 ms per 10^6 packed SNPs)
    ctx.scan_set_model(prep["A"], prep["w"], 0)
    rows = min(M, 250000)                       # a quarter of C3: 1.25 GB int8 / 5 GB fp32 of host memory
    host8 = g.download(0, rows)
    g2 = ctx.geno(M=rows, N=N)
    reps = 3

    def timed_upload(buf):
        g2.upload(buf)                          # warm-up (first touch of the pageable pages)
        ts = []
        for _ in range(reps):
            t0 = time.time()
            g2.upload(buf)
            ts.append(time.time() - t0)
        return float(np.median(ts))
    t8 = timed_upload(host8)
    host32 = host8.astype(np.float32)
    t32 = timed_upload(host32)
    same = np.array_equal(g2.download(), host8)
    g2.close()
    # pipelined: chunks uploaded on a second stream by the prefetch thread while the previous chunk is scanned
    # (the hdf5_data streaming loop, two stores ping-ponged), int8 host genotypes, ALL M rows
    host_all = g.download(0, M)
    src = {"c": {"raw_snps": host_all, "freqs": np.full(M, 0.5), "positions": np.arange(M)}}
    plan = hdf5_data._chunk_plan(src, 0.1, 50000)
    for _ci, _c, gg in hdf5_data._resident_chunks(ctx, src, plan[:2], reuse=True):     # allocate the two chunk stores
        ctx.scan(gg, prep["h0_rss"], n_p, fetch=True)
    t_pipes = []
    for _ in range(reps):
        t0 = time.time()
        for _ci, _c, gg in hdf5_data._resident_chunks(ctx, src, plan, reuse=True):
            ctx.scan(gg, prep["h0_rss"], n_p, fetch=True)
            gg.close()
        t_pipes.append(time.time() - t0)
    t_pipe = float(np.median(t_pipes))
    # packed rows (1 bit per 0/1 genotype, expanded on the device: mmg_geno_upload_packed): an eighth of the bytes
    from mixmogam_amd._lib import pack_genotypes
    packed_q = pack_genotypes(host8, 1)
    g3 = ctx.geno(M=rows, N=N)
    g3.upload_packed(packed_q, 1)
    tp = []
    for _ in range(reps):
        t0 = time.time()
        g3.upload_packed(packed_q, 1)
        tp.append(time.time() - t0)
    tpk = float(np.median(tp))
    same_packed = np.array_equal(g3.download(), host8)
    g3.close()
    packed_all = pack_genotypes(host_all, 1)
    del host_all
    hdf5_data.release_pools()
    srcp = {"c": {"raw_snps_packed": packed_all, "packed_bits": np.array(1), "num_indivs": np.array(N),
                  "freqs": np.full(M, 0.5), "positions": np.arange(M)}}
    t_pk = {}
    for csz, ramp in ((50000, False), (200000, False), (400000, True)):
        planp = hdf5_data._chunk_plan(srcp, 0.1, csz, ramp=ramp)
        pin = [ctx.pinned_empty(csz) for _ in range(3)]          # page-locked result buffers, reused by every chunk
        big2 = sorted(planp, key=lambda c: -len(c[1]))[:2]       # allocate the two chunk stores at full size
        for _ci, _c, gg in hdf5_data._resident_chunks(ctx, srcp, big2, reuse=True):
            ctx.scan(gg, prep["h0_rss"], n_p, fetch=True, out=[b[:gg.M] for b in pin])
        ts = []
        for _ in range(reps):
            t0 = time.time()
            for _ci, _c, gg in hdf5_data._resident_chunks(ctx, srcp, planp, reuse=True):
                ctx.scan(gg, prep["h0_rss"], n_p, fetch=True, out=[b[:gg.M] for b in pin])
                gg.close()
            ts.append(time.time() - t0)
        t_pk["%d%s" % (csz, "_ramp" if ramp else "")] = float(np.median(ts))
        hdf5_data.release_pools()
    del packed_all
    scan_s = ms_per_step * 1e-3 * rows / M
    best = min(t_pk, key=t_pk.get)
    return {"repeats": reps, "sample_rows": rows, "int8_upload_gbps": rows * N / t8 / 1e9, "f32_upload_convert_gbps": rows * N * 4.0 / t32 / 1e9,
            "packed1_upload_unpack_snps_per_s": rows / tpk, "packed1_upload_host_gbps": rows * (N / 8.0) / tpk / 1e9,
            "f32_ingest_round_trip_exact": bool(same), "packed1_ingest_round_trip_exact": bool(same_packed),
            "value_h2d_inclusive_int8_serial": rows / (t8 + scan_s),
            "value_h2d_inclusive_f32_serial": rows / (t32 + scan_s),
            "value_h2d_inclusive_packed1_serial": rows / (tpk + scan_s),
            "value_h2d_inclusive_int8_pipelined": M / t_pipe,
            "value_h2d_inclusive_packed1_pipelined": M / t_pk[best],
            "packed1_pipelined_by_chunk_size": {str(k): M / v for k, v in t_pk.items()},
            "packed1_pipelined_note": "<chunk>_ramp: first chunks of chunk/8, /4, /2 (hdf5_data._chunk_plan(ramp=True)) so "
                                      "that the first, unhidden upload is short",
            "pcie_roof_snps_per_s": {"int8": PCIE_GBPS * 1e9 / N, "f32": PCIE_GBPS * 1e9 / (4.0 * N),
                                     "packed1": PCIE_GBPS * 1e9 / (N / 8.0)},
            "note": "upload of the rows into the padded HBM store + the scan of those rows; serial = upload then "
                    "scan (sample_rows); pipelined = all M rows in chunks (50,000 SNPs for int8; see "
                    "packed1_pipelined_by_chunk_size) with the next upload overlapping "
                    "the current scan and the p-values of every chunk fetched; packed1 = 1 bit per 0/1 genotype on the "
                    "host, expanded on the device (mmg_geno_upload_packed)"}


FP64_PEAK_TFLOPS = 78.6      # MI355X_MICROARCH.md: fp64 vector = fp64 matrix (the two share the DP units)


def multi_roofline(N, M, q, batch, pass_ms, traffic=None):
    """Roofline of one pass of scan_multi over the rotated store: HBM (T is read once per pass, 8 N M bytes) against
    fp64 (2 + q fused multiply-adds per element and phenotype of the batch); the binding one is reported as `bound`."""
    hbm = 8.0 * N * M / (pass_ms * 1e-3) / 1e9
    tf = 2.0 * (2 + q) * batch * N * M / (pass_ms * 1e-3) / 1e12
    f_hbm, f_fp = hbm / HBM_PEAK_GBPS, tf / FP64_PEAK_TFLOPS
    common = {"kernel": "scan_multi_mfma_kernel" if batch >= 8 else "scan_multi_kernel", "ms": pass_ms,
              "phenotypes_per_pass": batch, "algorithmic_bytes": 8.0 * N * M,
              "algorithmic_flop": 2.0 * (2 + q) * batch * N * M,
              "hbm": {"achieved": hbm, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": f_hbm},
              "fp64": {"achieved": tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": f_fp,
                       "note": "sustained fp64 on this part, measured with tools/probe/fp64_rate.hip: 61-69 TFLOP/s on "
                               "the VALU, 36-48 on v_mfma_f64_16x16x4_f64, no more with both (shared units)"},
              "traffic": traffic, "traffic_from_profiles": True}
    if f_fp >= f_hbm:
        common.update({"bound": "fp64", "achieved": tf, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": f_fp})
    else:
        common.update({"bound": "hbm", "achieved": hbm, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": f_hbm})
    return common


def multi_record(ctx, g, lmm, N, M, P=16):
    """Multi-phenotype scans over the eigen-rotated store (mmg_rot_* / mmg_emmax_scan_multi), 16 random phenotypes."""
    from mixmogam_amd import linear_models as lm
    rng = np.random.RandomState(20260)
    ys = rng.standard_normal((P, N))
    eig_L = lmm._get_eigen_L_()
    t0 = time.time()
    models, d, omega, G = lm._multi_models(ys, lmm.X, eig_L)
    t_models = time.time() - t0
    try:
        rot = ctx.rot(eig_L["vectors"], M)
    except Exception as e:                       # 8 N M bytes of HBM
        return {"skipped": str(e)}
    rot.load(g)
    rot_ms = ctx.kernel_ms("rotate")
    h0 = np.array([m["h0_rss"] for m in models])
    ctx.scan_multi(rot, d, omega, G, h0, N - 2, want=("ps",))
    reps, walls, mss = 3, [], []
    for _ in range(reps):
        t0 = time.time()
        ctx.scan_multi(rot, d, omega, G, h0, N - 2, want=("ps",))
        walls.append(time.time() - t0)
        mss.append(ctx.kernel_ms("scan_multi"))
    wall, ms = float(np.median(walls)), float(np.median(mss))
    rot.close()
    from mixmogam_amd._lib import scan_multi_batch
    npass = -(-P // scan_multi_batch())
    return {"repeats": reps, "phenotypes": P, "phenotypes_per_pass": scan_multi_batch(), "rotation_gemm_ms": rot_ms,
            "rotation_executed_int8_tops": 2.0 * 4 * (-(-N // 256) * 256) * (-(-N // 64) * 64) * M / (rot_ms * 1e-3) / 1e12,
            "pass_ms": ms / npass, "passes": npass, "ms_per_phenotype_scan": ms / P,
            "value_snp_phenotype_scans_per_s": M * P / (wall), "value_kernels_only": M * P / (ms * 1e-3),
            "roofline": multi_roofline(N, M, 1, min(P, scan_multi_batch()), ms / npass,
                                       _profiled_traffic(N, M, 0, True).get("scan_multi_mfma_kernel")),
            "rotation_traffic_from_profiles": _profiled_traffic(N, M, 0, True).get("rot_gemm_w4_kernel"),
            "host_model_ms_per_phenotype": 1e3 * t_models / P}


# ----------------------------------------------------------------------------------------------- C4: permutation test
def perm_model_and_plan(ctx, lm, y, K_or_lmm, idx):
    """REML, scan model (loaded into the context) and permutation plan of hdf5_data.run_emmax_perm's flow without an
    eigendecomposition: the variance ratio from the band reduction of K, H_sqrt_inv := L^-1 of K + delta I = L L' where the
    scan model left it in HBM (linear_models.perm_h_from_cholesky).  Returns (prep, pp, plan, reml) -- the caller closes reml."""
    from scipy import linalg as _la
    if isinstance(K_or_lmm, lm.LinearMixedModel):
        lmm = K_or_lmm
    else:
        lmm = lm.LinearMixedModel(y, ctx=ctx)
        lmm.add_random_effect(K_or_lmm)
    t = [time.time()]
    lap = {}

    def mark(k):
        now = time.time(); lap[k] = now - t[0]; t[0] = now
    est = lmm.get_estimates_eigen_free()
    mark("reml")
    reml, delta = est["reml"], est["delta"]
    prep = lmm.scan_model_eigen_free(est)
    mark("scan_model")
    prep["delta"] = delta
    lmm_p = lm.LinearMixedModel(y, ctx=ctx)                               # perm_prepare centres Y in place (:1140)
    lmm_p.random_effects = lmm.random_effects
    pp = lmm_p.perm_prepare(None, num_perm=len(idx), perm_idx=idx, reml=reml, delta=delta)
    mark("perm_prepare")
    plan = reml.perm_plan(delta, pp["Ys"], pp["h0_rss"])
    mark("plan")
    Qc = _la.qr(pp["h0_X"], mode="economic")[0]
    prep["HtQ"] = np.ascontiguousarray(reml.linv_apply(delta, Qc, trans=True).T)
    mark("HtQ")
    prep["laps"] = lap
    return prep, pp, plan, reml


def bench_perm(args, ctx, coll, comm_h, g, lmm, y, N, M, Mtot, barrier, common):
    from mixmogam_amd import kinship, linear_models as lm
    P = args.perms
    idx = np.array([np.random.RandomState(20242 + p).permutation(N) for p in range(P)])
    # the SNP-independent half (REML, the scan model, H'H digit planes, W' = Ys'H digit image, v, Ys.Ys) is model setup
    perm_model_and_plan(ctx, lm, y, lmm, idx)[3].close()                  # (the first call of a process loads code objects)
    t0 = time.time()
    prep, pp, plan, reml = perm_model_and_plan(ctx, lm, y, lmm, idx)
    plan_s = time.time() - t0
    H = reml.linv(prep["delta"]) if (not args.no_cpu_baseline and common["n_gpus"] == 1) else None
    reml.close()
    Ys, h0_rss = pp["Ys"], pp["h0_rss"]

    def step():
        return plan.run(g, comm=comm_h)                                    # min over the SNP blocks of ALL ranks, in HBM

    for _ in range(args.warmup):
        step()
    perm_ms, tt_ms = [], []
    barrier()
    t0 = time.time()
    for _ in range(args.steps):
        min_rss = step()
        perm_ms.append(ctx.kernel_ms("perm"))
        try:
            tt_ms.append(ctx.kernel_ms("scan_quad"))                      # the t.t quadratic-form sweep of the stand-alone test
        except Exception:
            pass
    barrier()
    elapsed = time.time() - t0
    per_rank = [float(np.mean(perm_ms))]
    if coll is not None:
        elapsed = float(coll.allreduce(np.array([elapsed]), "max")[0])
        per_rank = [float(v) for v in coll.allgather(np.array([np.mean(perm_ms)]))]
        # every rank holds the same minima
        chk = coll.allreduce(min_rss.copy(), "max")
        if not np.array_equal(chk, min_rss):
            raise SystemExit("ranks disagree on the permutation minima")
    # the flow of hdf5_data.run_emmax_perm: the scan of the SNPs, then the test rebuilt from the scan's quadratic forms
    ctx.scan(g, prep["h0_rss"], prep["n_p"], fetch=False)
    plan.run(g, comm=comm_h, after_scan_HtQ=prep["HtQ"])
    barrier()
    t0 = time.time()
    for _ in range(args.steps):
        ctx.scan(g, prep["h0_rss"], prep["n_p"], fetch=False)
        min_rss_fast = plan.run(g, comm=comm_h, after_scan_HtQ=prep["HtQ"])
    barrier()
    fast_elapsed = time.time() - t0
    if coll is not None:
        fast_elapsed = float(coll.allreduce(np.array([fast_elapsed]), "max")[0])
    # ---- the whole job, kinship -> 5 % threshold, on the resident genotypes: IBS counts of every rank's SNP
    # block summed over RCCL and scaled on the device, REML + scan model + plan (replicated), scan + after-scan test of this
    # rank's block with the RCCL MAX of the statistics, thresholds on the host
    def whole_job():
        t0 = time.time()
        K2 = kinship.calc_ibs_kinship(None, geno=g, ctx=ctx, keep_device=True, comm=comm_h, m_total=Mtot)   # stays in HBM
        tk = time.time()
        prep2, pp2, plan2, reml2 = perm_model_and_plan(ctx, lm, y, K2, idx)
        reml2.close()
        K2.close()
        tm = time.time()
        ps = ctx.scan(g, prep2["h0_rss"], prep2["n_p"])["ps"]
        mr = plan2.run(g, comm=comm_h, after_scan_HtQ=prep2["HtQ"])
        plan2.close()
        mf = (pp2["h0_rss"] / mr - 1.0) * pp2["n_p"]
        mp = ctx.f_sf(mf, pp2["n_p"])
        thr = float(np.sort(mp)[P // 20])
        t1 = time.time()
        ph = {"kinship": tk - t0, "reml_model_plan": tm - tk, "scan_test_threshold": t1 - tm}
        ph.update({"..." + k: v for k, v in prep2["laps"].items()})
        return t1 - t0, ph, thr, float(ps.min())
    whole_job()
    barrier()
    e2e = [whole_job() for _ in range(2)]
    barrier()
    e2e_s, e2e_phases, e2e_thr, _mp = min(e2e, key=lambda v: v[0])
    if coll is not None:
        e2e_s = float(coll.allreduce(np.array([e2e_s]), "max")[0])
    if (coll.rank if coll is not None else 0) != 0:
        return None
    n_p = N - 2
    max_f = (h0_rss / min_rss - 1.0) * n_p
    min_ps = ctx.f_sf(max_f, n_p)
    Npad = -(-N // 256) * 256
    Ppad = -(-P // 64) * 64
    ex = 2.0 * 256.0 * 256.0 * Npad * (Ppad / 64) * (-(-M // 256))       # 256 x 256 tiles (64 permutations x 4 digit rows), int8 ops
    res = dict(common)
    res.update({"metric": "SNP x permutation tests/sec, EMMAX permutation test (BASELINE configs[3])",
                "value": float(Mtot) * P * args.steps / elapsed, "unit": "SNP-permutations/s",
                "ms_per_step": 1e3 * elapsed / args.steps, "scaling": "strong", "dtype": "i8",
                "config": {"workload": "N=%d x M=%d SNPs in total, P=%d permutations, SNP blocks sharded over the ranks, "
                                       "RCCL MAX all-reduce of the P statistics in HBM inside the timed region; the "
                                       "SNP-independent operand images (mmg_perm_plan_create) are setup"
                                       % (N, Mtot, P), "snps_per_gpu": M, "parallelism": "snp-block x%d" % common["n_gpus"]},
                "roofline": {"bound": "mfma", "kernel": "perm_gemm_w4_kernel", "ms": per_rank[0],
                             "achieved": 2.0 * N * P * M / (per_rank[0] * 1e-3) / 1e12, "peak": I8_MFMA_PEAK_TOPS,
                             "unit": "TFLOP/s", "frac": 2.0 * N * P * M / (per_rank[0] * 1e-3) / 1e12 / I8_MFMA_PEAK_TOPS,
                             "executed_int8_tops": ex / (per_rank[0] * 1e-3) / 1e12,
                             "executed_frac": ex / (per_rank[0] * 1e-3) / 1e12 / I8_MFMA_PEAK_TOPS},
                # two thirds of a stand-alone test is not the permutation GEMM but the quadratic form t.t = s'(C A' C)s of every
                # SNP -- a scan GEMM of its own over the centred model (adaptive digit schedule); the chunked driver's flow
                # below reuses the scan's quadratic forms instead (scan_plus_test_after_scan)
                "roofline_tt_sweep": ({"bound": "mfma", "kernel": QUAD_KERNEL, "ms": float(np.mean(tt_ms)),
                                       "achieved": (2.0 * N * N + 4.0 * N) * M / (float(np.mean(tt_ms)) * 1e-3) / 1e12,
                                       "peak": I8_MFMA_PEAK_TOPS, "unit": "TFLOP/s",
                                       "frac": (2.0 * N * N + 4.0 * N) * M / (float(np.mean(tt_ms)) * 1e-3) / 1e12 / I8_MFMA_PEAK_TOPS,
                                       "note": "the same kernel and algorithmic work as the headline scan (roofline of --mode weak)"}
                                      if tt_ms else None),
                "perm_gemm_ms_per_rank": per_rank, "model_and_plan_s": plan_s,
                "end_to_end_perm_s": e2e_s, "end_to_end_perm_phases_s": e2e_phases, "end_to_end_threshold_05_min_p": e2e_thr,
                "end_to_end_s": e2e_s, "end_to_end_phases_s": e2e_phases,          # the keys every mode carries
                "end_to_end_note": "kinship (IBS counts of the resident SNPs, RCCL sum, scale_k on the device, kept in HBM) -> REML "
                                   "(band reduction) -> scan model + permutation plan from the Cholesky factor in HBM "
                                   "(H_sqrt_inv := L^-1, no eigendecomposition) -> EMMAX scan + after-scan permutation test "
                                   "-> 5 % threshold on the host; best of two warm runs, max over the ranks",
                "scan_plus_test_after_scan": {
                    "ms_per_step": 1e3 * fast_elapsed / args.steps,
                    "note": "EMMAX scan of the SNPs + permutation test rebuilt from the scan's quadratic forms "
                            "(mmg_emmax_perm_after_scan): the per-chunk work of hdf5_data.run_emmax_perm",
                    "max_rel_min_rss_diff_vs_standalone": float(np.max(np.abs(min_rss_fast / min_rss - 1)))},
                "threshold_05": {"min_p": float(np.sort(min_ps)[P // 20]), "max_f": float(np.sort(max_f)[::-1][P // 20])}})
    if not args.no_cpu_baseline and common["n_gpus"] == 1:
        res["cpu_baseline"] = cpu_baseline_perm(N, M, P, H, Ys, h0_rss, g, min_rss)
    return res


def cpu_baseline_perm(N, M, P, H, Ys, h0_rss, g, gpu_min_rss):
    """the C4 CPU baseline on a 1/100 SNP subsample, extrapolated linearly in M.  What is timed is the
    arithmetic of linear_models.py:1157-1164 as the oracle restates it in closed form (perm_closed: row-centred SNPs,
    t = H s, per-SNP rss over all P permutations as two float64 GEMMs) -- the reference's own loop runs one multi-RHS
    lstsq per SNP on top of that, so this is a LOWER bound of its time."""
    from oracle import emmax_oracle as orc
    sample = max(N, M // 100)
    snps = g.download(0, sample)
    pp = {"H": H, "Ys": Ys, "h0_rss": h0_rss, "yy": np.einsum("ij,ij->j", Ys, Ys), "n": N, "q": 1}
    t0 = time.time()
    out = orc.perm_closed(snps, pp)
    dt = time.time() - t0
    # the minimum over a subsample is >= the minimum over all SNPs
    ok = bool(np.all(out["min_rss"] >= gpu_min_rss * (1 - 1e-9)))
    return {"value": float(sample) * P / dt, "unit": "SNP-permutations/s", "cores": os.cpu_count(), "kind": "port",
            "sample": "first %d SNPs (1/100 of the workload; linear in M), closed-form float64 restatement "
                      "(oracle.perm_closed), %.1f s" % (sample, dt),
            "subsample_minima_bound_the_gpu_minima": ok}


# ----------------------------------------------------------------------------------------------- multi-phenotype
def bench_multi(args, ctx, coll, g, lmm, eig_L, N, M, Mtot, barrier, common):
    from mixmogam_amd import linear_models as lm
    P = args.phenos
    rng = np.random.RandomState(20260)
    ys = rng.standard_normal((P, N))
    t0 = time.time()
    models, d, omega, G = lm._multi_models(ys, lmm.X, eig_L)
    t_models = time.time() - t0
    rot = ctx.rot(eig_L["vectors"], M)
    rot.load(g)
    rot_ms = ctx.kernel_ms("rotate")
    h0 = np.array([m["h0_rss"] for m in models])
    pinned = {"ps": ctx.pinned_empty(P * M)}                 # results land in page-locked memory, reused by every step
    for _ in range(args.warmup):
        ctx.scan_multi(rot, d, omega, G, h0, N - 2, want=("ps",), out=pinned)
    ms = []
    barrier()
    t0 = time.time()
    for _ in range(args.steps):
        out = ctx.scan_multi(rot, d, omega, G, h0, N - 2, want=("ps",), out=pinned)   # P x M p-values on this rank's host
        ms.append(ctx.kernel_ms("scan_multi"))
    barrier()
    elapsed = time.time() - t0
    if coll is not None:
        elapsed = float(coll.allreduce(np.array([elapsed]), "max")[0])
    rot.close()
    # the whole multi-phenotype job on this rank as one clock: null models of all phenotypes (host), rotation of the resident
    # SNPs into the eigenbasis, every phenotype's pass, p-values on the host; max over the ranks
    def whole_job():
        t0 = time.time()
        mo, d2, om2, G2 = lm._multi_models(ys, lmm.X, eig_L)
        t1 = time.time()
        r2 = ctx.rot(eig_L["vectors"], M)
        r2.load(g)
        t2 = time.time()
        ctx.scan_multi(r2, d2, om2, G2, np.array([m["h0_rss"] for m in mo]), N - 2, want=("ps",), out=pinned)
        t3 = time.time()
        r2.close()
        return t3 - t0, {"null_models_host": t1 - t0, "rotation": t2 - t1, "scan_passes": t3 - t2}
    barrier()
    job_s, job_ph = min((whole_job() for _ in range(2)), key=lambda v: v[0])
    barrier()
    if coll is not None:
        job_s = float(coll.allreduce(np.array([job_s]), "max")[0])
    if (coll.rank if coll is not None else 0) != 0:
        return None
    from mixmogam_amd._lib import scan_multi_batch
    npass = -(-P // scan_multi_batch())
    pass_ms = float(np.mean(ms)) / npass
    res = dict(common)
    res.update({"metric": "SNP x phenotype EMMAX scans/sec over the eigen-rotated store",
                "value": float(Mtot) * P * args.steps / elapsed, "unit": "SNP-phenotype scans/s",
                "ms_per_step": 1e3 * elapsed / args.steps, "scaling": "weak", "dtype": "f64",
                "config": {"workload": "N=%d x M=%d SNPs per GPU, %d phenotypes with their own delta, %d per pass" % (N, M, P, scan_multi_batch()),
                           "parallelism": "snp-block x%d" % common["n_gpus"]},
                # (a pass moves the same bytes whatever the phenotype count of the run: the committed per-launch figure applies as
                # long as the pass has the profile's batch -- round 4 shipped `traffic: null` for the 64-phenotype default)
                "roofline": multi_roofline(N, M, 1, min(P, scan_multi_batch()), pass_ms,
                                           _profiled_traffic(N, M, 0, True).get("scan_multi_mfma_kernel")
                                           if min(P, scan_multi_batch()) == 16 else None),
                "rotation_gemm_ms": rot_ms, "host_model_ms_per_phenotype": 1e3 * t_models / P,
                "end_to_end_s": job_s, "end_to_end_phases_s": job_ph,
                "min_p": float(out["ps"].min())})
    return res


def bench_c5(args, ctx, coll, rank, world, info, rccl_nranks, identity=None, rccl=None):
    """BASELINE config 5 as a job: N = 50,000 individuals x M = 10,000,000 SNPs streamed chunk-wise, the chunks dealt
    round-robin over the ranks -- hdf5_data.run_emmax (hdf5_data.py:70-187 of the reference) on a lazily generated,
    1-bit packed genotype tree (simulations.lazy_synthetic_source: every chunk is regenerated on each read, the 500 GB
    matrix exists nowhere), kinship partial sums all-reduced in HBM, REML + scan model replicated, scan results
    all-gathered.  A step is the whole pipeline; the line carries the stage times of every rank.  On one GPU
    --c5-share-of W times the share one rank of W would own (no peers: its kinship is that share's)."""
    from mixmogam_amd import hdf5_data, simulations
    N, share_of = args.c5_n, args.c5_share_of
    if share_of and world > 1:
        raise SystemExit("--c5-share-of is the one-GPU stand-in for a multi-rank run")
    Mtot = args.c5_m // share_of if share_of else args.c5_m
    tree, y = simulations.lazy_synthetic_source(N, Mtot, num_chroms=5, gen_rows=6250, seed=20240, pheno_seed=20241,
                                                num_causals=100, threads=int(os.environ.get("MMG_BENCH_GEN_THREADS", "8")),
                                                packed=True)

    def barrier():                                             # every library call blocks on its HIP stream: the device is idle here
        if coll is not None:
            coll.barrier()

    stage_keys = ("kinship_pass_s", "grm_kernel_s", "reml_s", "scan_model_s", "scan_pass_s", "scan_kernel_s", "gather_s")
    out = T = None
    for _ in range(args.warmup):
        hdf5_data.run_emmax(tree, y, min_maf=0.1, chunk_size=args.c5_chunk, ctx=ctx, coll=coll)
    barrier()
    t0 = time.time()
    for _ in range(args.steps):
        T = {}
        out = hdf5_data.run_emmax(tree, y, min_maf=0.1, chunk_size=args.c5_chunk, ctx=ctx, coll=coll, timings=T)
    barrier()
    dt = time.time() - t0
    mine = np.array([dt] + [float(T.get(k, 0.0)) for k in stage_keys])
    if coll is not None:
        allv = np.asarray(coll.allgather(mine)).reshape(world, len(mine))
    else:
        allv = mine[None, :]
    dt_max = float(allv[:, 0].max())
    if rank != 0:
        return
    ps = np.concatenate([out["chrom_results"][c]["ps"] for c in out["chrom_results"]])
    n_snps = int(out["num_snps"])
    nt = -(-N // 256)
    grm_ops = 4.0 * 2.0 * 256.0 * 256.0 * (nt * (nt + 1) / 2)          # per SNP: 4 planes over the 256-tile triangle
    grm_s = float(allv[:, 2].sum())                                      # all ranks' GEMM seconds for all SNPs
    rec = {"metric": "SNPs/sec EMMAX end to end (kinship + REML + scan, streamed)", "unit": "SNPs/s",
           "value": n_snps * args.steps / dt_max, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": 1e3 * dt_max / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
           "dtype": "i8", "data": "synthetic", "rccl_nranks": rccl_nranks, "mode": "c5", "devices": identity, "rccl": rccl,
           "rccl_allreduce_gbps": rccl["allreduce_busbw_gbps"] if rccl else None,
           # the step IS the whole job here: the same keys the other modes carry
           "end_to_end_s": dt_max / args.steps,
           "end_to_end_phases_s": {k: float(allv[:, 1 + i].max()) for i, k in enumerate(stage_keys)},
           "setup_s_per_rank": [0.0] * world,
           "config": {"workload": "BASELINE configs[4]: N=%d individuals x M=%d SNPs streamed in chunks of %d from a lazily "
                                  "generated 1-bit packed source%s" % (N, Mtot, args.c5_chunk,
                                  " -- the share of ONE rank of %d, timed alone on one GPU" % share_of if share_of else ""),
                      "n_individuals": N, "snps_total": n_snps, "chunk": args.c5_chunk, "share_of": share_of or None,
                      "parallelism": "chunks round-robin x%d; kinship all-reduce in HBM, REML + scan model replicated" % world},
           "stage_s_per_rank": {k: [round(float(v), 3) for v in allv[:, 1 + i]] for i, k in enumerate(stage_keys)},
           "route": T.get("route"),
           "roofline": {"bound": "mfma", "kernel": "kinship_grm4_kernel", "unit": "TOP/s", "peak": 5000.0,
                        "achieved": None, "frac": grm_ops * n_snps / max(grm_s, 1e-9) / 1e12 / 5000.0,
                        "note": "executed int8 ops of the kinship pass (its GEMM kernels are the largest stage) over their "
                                "summed kernel time on all ranks; the scan's kernel time is scan_kernel_s", "traffic": None},
           "cpu_baseline": (cpu_baseline_c5(N) if (not args.no_cpu_baseline and world == 1) else None),
           "pseudo_heritability": float(out["pseudo_heritability"]), "min_p": float(ps.min()),
           "n_p_below_1e-8": int((ps < 1e-8).sum()), "device": info}
    rec["roofline"]["achieved"] = rec["roofline"]["frac"] * 5000.0
    print(json.dumps(rec))


def cpu_baseline_c5(N, sample=1024):
    """the CPU path on a SUBSAMPLE, extrapolated linearly in M.  At N = 50,000 the reference's two
    per-SNP stages are timed on `sample` SNPs each, in the shapes and dtypes its loops use: the kinship chunk product
    (2S - 1)(2S - 1)' in float64 (kinship.py:29-44) and the scan chunk -- float32 `chunk @ M` with an N x N matrix, one
    scipy.linalg.lstsq per SNP, f.sf (linear_models.py:1315-1349).  The projection matrix is a constant-filled stand-in (BLAS
    time does not depend on the values; the real one needs eigh(K) of a 50,000 x 50,000 matrix, ~1e15 flop, which is NOT
    timed and NOT in the figure: the baseline is an upper bound of the CPU rate)."""
    import scipy.linalg
    import scipy.stats
    rng = np.random.RandomState(5)
    chunk = (rng.random_sample((sample, N)) < 0.5).astype(np.int8)
    t0 = time.time()
    x = (2.0 * chunk - 1.0)                                              # float64, as sp.mat(..., dtype='single') * 2 - 1 promotes
    k_part = x.T @ x
    t_kin = time.time() - t0
    del k_part, x
    Mp = np.full((N, N), 1e-3, dtype=np.float32)
    r = rng.standard_normal(N).astype(np.float32)
    t0 = time.time()
    Xs = chunk.astype(np.float32) @ Mp                                   # :1317-1318
    rss = np.empty(sample)
    for j in range(sample):                                              # :1328
        rss[j] = float(np.sum((r - Xs[j] * scipy.linalg.lstsq(Xs[j][:, None], r)[0][0]) ** 2))
    scipy.stats.f.sf(np.maximum(rss, 1e-30), 1, N - 2)
    t_scan = time.time() - t0
    per_snp = (t_kin + t_scan) / sample
    return {"value": 1.0 / per_snp, "unit": "SNPs/s", "cores": os.cpu_count(), "kind": "port",
            "sample": "%d SNPs at N = %d through the reference's two per-SNP stages (kinship chunk product in float64 %.1f s, "
                      "scan chunk float32 GEMM + one lstsq per SNP + f.sf %.1f s), extrapolated linearly in M; eigh(K) "
                      "(~1e15 flop at this N) is not timed and not in the figure" % (sample, N, t_kin, t_scan)}


def _device_rows(ctx, rows, n, seed):
    """Genotype rows of arbitrary global SNP indices, regenerated on the device by the counter hash."""
    tmp = ctx.geno(M=1, N=n)
    out = []
    for r in rows:
        tmp.fill_hash(seed, m_global0=int(r))
        out.append(tmp.download()[0])
    tmp.close()
    return np.vstack(out)


def cpu_baseline(N, sample, lmm, est, prep, gpu_ps, g):
    """The reference's loop structure (chunks of N SNPs, float32 `chunk @ M` GEMM, one
    scipy.linalg.lstsq per SNP, scipy.stats.f.sf -- linear_models.py:1315-1349) as restated in
    oracle/emmax_oracle.py:scan_loop, timed on this box's host cores over the first `sample` SNPs of the same
    workload (default: all of them).  The genotype rows come from the device store block by block (untimed; the
    first block is checked against the oracle's own host generator, the two share the counter hash).  Reported,
    not a target."""
    from scipy import linalg
    from oracle import emmax_oracle as orc
    H = np.asarray(est["H_sqrt_inv"])
    h0_X = H @ lmm.X
    Q, _ = linalg.qr(h0_X, mode="economic")
    Mp = (H - Q @ (Q.T @ H)).T                                   # H'(I - QQ'), O(N^2 q)
    p = {"n": N, "q": lmm.X.shape[1], "Mp": Mp, "r": prep["r"], "h0_rss": prep["h0_rss"], "h0_betas": prep["h0_betas"]}
    block = 20 * N                                               # 20 of the reference's chunks per download
    dt, agree, first = 0.0, 0.0, None
    for c0 in range(0, sample, block):
        rows = min(block, sample - c0)
        snps = g.download(c0, rows)
        if first is None:
            first = snps[:N].copy()
            if not np.array_equal(first, orc.hash_genotypes(0, len(first), N, 20240)):
                raise SystemExit("device genotypes differ from the oracle's host generator")
        t0 = time.time()
        out = orc.scan_loop(snps, p, dtype=np.float32)
        dt += time.time() - t0
        agree = max(agree, float(np.nanmax(np.abs(out["ps"] / gpu_ps[c0:c0 + rows] - 1))))
    blas, blas4 = "unknown", None
    try:
        import threadpoolctl
        blas = ";".join("%s:%s" % (d.get("internal_api"), d.get("num_threads")) for d in threadpoolctl.threadpool_info())
        # the reference pins its BLAS to 4 threads (linear_models.py:37): the same loop on a bounded sample
        n4 = min(sample, 20 * N)
        snps = g.download(0, n4)
        with threadpoolctl.threadpool_limits(limits=4):
            t0 = time.time()
            orc.scan_loop(snps, p, dtype=np.float32)
            blas4 = n4 / (time.time() - t0)
    except Exception:
        pass
    return {"value": sample / dt, "value_blas_4_threads": blas4, "unit": "SNPs/s", "cores": os.cpu_count(), "kind": "port",
            "sample": "%s %d SNPs of the same workload (chunks of N), fp32 reference loop, %.1f s; BLAS %s; "
                      "4-thread figure on the first %d SNPs"
                      % ("all" if sample == g.M else "first", sample, dt, blas, min(sample, 20 * N)),
            "max_rel_p_diff_vs_gpu": agree}


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Second randomised sweep: the paths of random_parity.py's blind spots -- larger N on the band route with duplicated and
closely related individuals, the public permutation test, with_betas, the exact-EMMA refinement, and the chunked driver
(random chunk sizes, container vs in-memory) -- against the float64 oracle.  Checker only.
    python tools/random_parity2.py [cases] [seed]"""
import os, sys, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mixmogam_amd import _lib, kinship, hdf5_data, linear_models as lm
from oracle import emmax_oracle as orc
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = _lib.get_context()
rng = np.random.RandomState(seed)
worst, fails = {}, 0


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))) if a.size else 0.0


def note(key, val, tol, what):
    global fails
    worst[key] = max(worst.get(key, 0.0), val)
    if not val <= tol:
        fails += 1
        print("  FAIL %-22s %.3e > %.1e   %s" % (key, val, tol, what), flush=True)


for c in range(cases):
    n = int(rng.choice([rng.randint(20, 250), rng.randint(256, 900), rng.randint(900, 2600)]))
    m = int(rng.randint(50, 4000))
    freq = rng.uniform(0.05, 0.95, m)
    base = (rng.random_sample((m, n)) < freq[:, None]).astype(np.int8)
    kind = rng.choice(["plain", "duplicates", "relatives", "few classes"])
    if kind == "duplicates":                                   # a third of the individuals are exact copies of others
        src = rng.randint(0, n, n // 3); dst = rng.choice(n, n // 3, replace=False)
        base[:, dst] = base[:, src]
    elif kind == "relatives":                                  # pairs that differ in 1 % of the SNPs
        for a in range(0, n - 1, 2):
            base[:, a + 1] = base[:, a]
            flip = rng.random_sample(m) < 0.01
            base[flip, a + 1] ^= 1
    elif kind == "few classes":                                # every individual is one of 12 genotype vectors
        proto = base[:, :12]
        base = proto[:, rng.randint(0, 12, n)]
    snps = base[base.std(1) > 0]
    if len(snps) < 3:
        continue
    m = len(snps)
    y = rng.standard_normal(n) + 0.8 * snps[rng.randint(m)] + 0.5 * snps[rng.randint(m)]
    what = "case %d: n=%d m=%d %s" % (c, n, m, kind)
    if os.environ.get("RP_VERBOSE"):
        print(what, flush=True)
    try:
        K = kinship.calc_ibs_kinship(snps, ctx=ctx)
        note("ibs K", float(np.max(np.abs(K - orc.calc_ibs_kinship(snps)))), 1e-12, what)
        os.environ.get("RP_VERBOSE") and print("   ->", 'orc.emmax', flush=True)
        ref = orc.emmax(snps, y, K)
        ok = ref["ps"] > 1e-290
        res = lm.emmax(snps, list(y), K, ctx=ctx)
        note("emmax p", rel(res["ps"][ok], ref["ps"][ok]), 1e-6, what + " " + str(ctx.scan_last_stats()))
        note("emmax h2", abs(res["pseudo_heritability"] - ref["pseudo_heritability"]), 1e-6, what)
        if rng.rand() < 0.5:
            os.environ.get("RP_VERBOSE") and print("   ->", 'lm.emmax', flush=True)
            wb = lm.emmax(snps, list(y), K, with_betas=True, ctx=ctx)
            note("with_betas p", rel(wb["ps"][ok], ref["ps"][ok]), 1e-6, what)
        if rng.rand() < 0.5 and n <= 900 and kind != "few classes":   # (identical SNPs tie: which of them gets refined is arbitrary)
            en = int(rng.randint(1, 6))
            # which SNPs get the exact test is decided by the order of the EMMAX p-values: two that differ by less than the
            # parity bar (a SNP and its complement, duplicates up to rounding -- common at small n) may swap places between
            # device and oracle, and then different SNPs are refined.  Such a near-tie AT the cutoff is not a parity question.
            ps_sorted = np.sort(ref["ps"])
            if len(ps_sorted) > en and ps_sorted[en] - ps_sorted[en - 1] <= 1e-5 * ps_sorted[en]:
                en = 0
            os.environ.get("RP_VERBOSE") and print("   ->", 'lm.emmax', flush=True)
            if en:
                em = lm.emmax(snps, list(y), K, emma_num=en, ctx=ctx)
                eo = orc.emmax_with_emma(snps, y, K, emma_num=en)
                oke = eo["ps"] > 1e-290
                note("emma_num p", rel(em["ps"][oke], eo["ps"][oke]), 2e-6, what + " emma_num=%d" % en)
        if rng.rand() < 0.5 and n <= 900:
            P = int(rng.randint(2, 40))
            idx = np.array([rng.permutation(n) for _ in range(P)])
            est = orc.get_estimates(y, np.ones((n, 1)), orc.scale_k(K))
            os.environ.get("RP_VERBOSE") and print("   ->", 'lm.emmax_perm_test', flush=True)
            got = lm.emmax_perm_test(snps, list(y), K, num_perm=P, perm_idx=idx, H_sqrt_inv=est["H_sqrt_inv"], ctx=ctx)
            want = orc.perm_public(snps, y, np.ones((n, 1)), est["H_sqrt_inv"], idx, reference_indexing=False)
            note("perm max F", rel(got["max_f_stats"], want["max_f_stats"]), 1e-6, what + " P=%d" % P)
        if rng.rand() < 0.5 and n <= 900:
            ys = np.vstack([y, rng.standard_normal(n) + snps[rng.randint(m)], rng.standard_normal(n)])
            os.environ.get("RP_VERBOSE") and print("   -> emmax_multi", flush=True)
            mr = lm.emmax_multi(snps, ys, K, ctx=ctx)
            mo = orc.emmax_multi(snps, ys, K)
            # a phenotype without a genetic component has a FLAT likelihood at its optimum (delta ~ 1e3: h2 ~ 0), and where the
            # secant search stops on it is decided by rounding (the reference's own stopping rule, :847: |step| < 1.48e-8): two
            # correct searches end 1e-4 apart in delta with likelihoods equal to 1e-12, and p moves with delta like lambda_max / delta.
            # Such a phenotype is compared at the SAME delta: the oracle's scan with the variance ratio the product found.
            mo_ps = mo["ps"].copy()
            for pi in range(len(ys)):
                dp, do = float(mr["delta"][pi]), float(mo["delta"][pi])
                if abs(dp / do - 1) > 1e-7 and abs(float(mr["max_ll"][pi]) - float(mo["max_ll"][pi])) <= 1e-9 * abs(float(mo["max_ll"][pi])):
                    lam_k, U_k = np.linalg.eigh(orc.scale_k(K))
                    Hp = (1.0 / np.sqrt(lam_k + dp))[:, None] * U_k.T
                    mo_ps[pi] = orc.scan_closed(snps, orc.scan_prepare(ys[pi], np.ones((n, 1)), Hp))["ps"]
                    note("emmax_multi flat-optimum delta", abs(dp / do - 1), 1e-2, what + " phenotype %d: delta %g vs %g" % (pi, dp, do))
            okm = mo_ps > 1e-290
            note("emmax_multi p", rel(np.asarray(mr["ps"])[okm], mo_ps[okm]), 1e-6, what)
            if os.environ.get("RP_DEBUG_MULTI") and rel(np.asarray(mr["ps"])[okm], mo_ps[okm]) > 1e-6:
                lam = np.linalg.eigvalsh(orc.scale_k(K))
                e = np.abs(np.asarray(mr["ps"]) / mo_ps - 1)
                print("   multi debug: delta", np.asarray(mr["delta"]), "oracle delta", mo["delta"], "max_ll", mr["max_ll"], mo["max_ll"], "lambda quantiles", np.quantile(lam, [0, .01, .1, .5, .9, 1]),
                      "bad SNPs per phenotype", (e > 1e-6).sum(1), "worst per phenotype", e.max(1), "p at worst", [float(mo["ps"][i, e[i].argmax()]) for i in range(len(e))], flush=True)
        if rng.rand() < 0.6:
            cuts = sorted(set([0, m] + list(rng.randint(1, m, 2))))
            tree = {"c%d" % i: {"raw_snps": snps[a:b], "freqs": snps[a:b].mean(1), "positions": np.arange(b - a)}
                    for i, (a, b) in enumerate(zip(cuts[:-1], cuts[1:]))}
            chunk = int(rng.choice([7, 64, 1000, 10 ** 6]))
            os.environ.get("RP_VERBOSE") and print("   ->", 'hdf5_data.run_emmax', flush=True)
            out = hdf5_data.run_emmax(tree, y, min_maf=0.0, chunk_size=chunk, ctx=ctx)
            Kg = orc.calc_ibd_kinship(snps)
            note("run_emmax kinship", float(np.max(np.abs(out["kinship"] - Kg)) / np.max(np.abs(Kg))), 2e-9, what)
            r2 = orc.emmax(snps, y, Kg)
            got = np.concatenate([out["chrom_results"][k]["ps"] for k in tree])
            ok2 = r2["ps"] > 1e-290
            note("run_emmax p", rel(got[ok2], r2["ps"][ok2]), 1e-6, what + " chunk=%d" % chunk)
    except Exception as e:                                     # noqa: report and continue -- that IS the finding
        fails += 1
        print("  EXCEPTION %s: %s: %s" % (what, type(e).__name__, str(e)[:300]), flush=True)
        traceback.print_exc(limit=4)
print("worst over %d cases: %s" % (cases, {k: "%.2e" % v for k, v in worst.items()}))
print("failures: %d" % fails)
sys.exit(1 if fails else 0)

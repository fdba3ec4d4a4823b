#!/usr/bin/env python3
"""Fourth randomised sweep: the file paths and the replicate design -- genotype containers (int8 rows, 1-bit and 2-bit packed
rows) of random shape through hdf5_data.run_emmax / run_emmax_perm / calculate_ibd_kinship with random chunk sizes and MAF
filters against the same data in memory and against the oracle; emmax() with an incidence matrix Z of replicated
measurements.  Checker only.   python tools/random_parity4.py [cases] [seed]"""
import os, shutil, sys, tempfile, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mixmogam_amd import _lib, chunkstore, hdf5_data, kinship, linear_models as lm
from oracle import emmax_oracle as orc
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = _lib.get_context()
rng = np.random.RandomState(seed)
worst, fails = {}, 0
tmp = tempfile.mkdtemp(prefix="mmg_rp4_")


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))) if a.size else 0.0


def note(key, val, tol, what):
    global fails
    worst[key] = max(worst.get(key, 0.0), val)
    if not val <= tol:
        fails += 1
        print("  FAIL %-24s %.3e > %.1e   %s" % (key, val, tol, what), flush=True)


try:
    for c in range(cases):
        n = int(rng.choice([rng.randint(12, 200), rng.randint(256, 900)]))
        alphabet = rng.choice(["binary", "diploid"])
        nchrom = int(rng.randint(1, 4))
        chroms = {}
        for ci in range(nchrom):
            mc = int(rng.randint(20, 1500))
            f = rng.uniform(0.02, 0.98, mc)
            s = (rng.random_sample((mc, n)) < f[:, None]).astype(np.int8)
            if alphabet == "diploid":
                s = s + (rng.random_sample((mc, n)) < f[:, None]).astype(np.int8)
            chroms["chr%d" % (ci + 1)] = s
        allsnps = np.vstack(list(chroms.values()))
        y = rng.standard_normal(n) + 0.7 * allsnps[rng.randint(len(allsnps))]
        bits = 0 if rng.rand() < 0.4 else (1 if alphabet == "binary" else 2)
        maf = float(rng.choice([0.0, 0.05, 0.1]))
        chunk = int(rng.choice([13, 200, 100000]))
        what = "case %d: n=%d chroms=%s %s packed_bits=%d maf=%.2f chunk=%d" % (c, n, [len(v) for v in chroms.values()], alphabet, bits, maf, chunk)
        try:
            path = os.path.join(tmp, "g%d" % c)
            chunkstore.write_genotype_container(path, chroms, np.arange(n), phenotypes=y, packed_bits=bits)
            out = hdf5_data.run_emmax(path, None, min_maf=maf, chunk_size=chunk, ctx=ctx)
            # the reference's MAF filter on `freqs` (hdf5_data.py:91-96) -- the allele frequency as its parser writes it: mean / 2 for
            # 0/1/2 codes (plink2hdf5.py:202), the carrier frequency for 0/1 codes (chunkstore.write_genotype_container's default)
            fr = {k: v.mean(1) / (2.0 if v.max() > 1 else 1.0) for k, v in chroms.items()}
            keep = {k: np.minimum(fr[k], 1 - fr[k]) > maf for k in chroms}
            kept = np.vstack([chroms[k][keep[k]] for k in chroms])
            poly = kept.std(1) > 0
            if not poly.all() or len(kept) < 3 or n - 2 < 3:
                continue                                            # (a monomorphic SNP in the GRM is an error in the reference too)
            Kg = orc.calc_ibd_kinship(kept)
            note("file kinship", float(np.max(np.abs(out["kinship"] - Kg)) / np.max(np.abs(Kg))), 5e-9, what)   # header: "entries good to ~1e-9"
            ref = orc.emmax(kept, y, Kg)
            got = np.concatenate([out["chrom_results"][k]["ps"] for k in chroms])
            ok = ref["ps"] > 1e-290
            note("file run_emmax p", rel(got[ok], ref["ps"][ok]), 1e-6, what)
            note("file h2", abs(out["pseudo_heritability"] - ref["pseudo_heritability"]), 1e-6, what)
            mem = {k: {"raw_snps": v, "freqs": fr[k], "positions": np.arange(len(v))} for k, v in chroms.items()}
            out2 = hdf5_data.run_emmax(mem, y, min_maf=maf, chunk_size=int(rng.choice([50, 100000])), ctx=ctx)
            got2 = np.concatenate([out2["chrom_results"][k]["ps"] for k in chroms])
            note("file vs memory p", rel(got, got2), 1e-7, what)
            if rng.rand() < 0.4 and n <= 400:
                P = int(rng.randint(3, 30))
                idx = np.array([rng.permutation(n) for _ in range(P)])
                op = hdf5_data.run_emmax_perm(path, None, min_maf=maf, chunk_size=chunk, num_perm=P, perm_idx=idx, ctx=ctx)
                om = hdf5_data.run_emmax_perm(mem, y, min_maf=maf, chunk_size=100000, num_perm=P, perm_idx=idx, ctx=ctx)
                note("perm file vs memory", rel(np.sort(op["perm_max_f_stats"]), np.sort(om["perm_max_f_stats"])), 1e-7, what + " P=%d" % P)
            # replicated measurements: n_values rows over the n individuals
            if rng.rand() < 0.5 and alphabet == "binary":
                nv = n + int(rng.randint(1, n))
                who = np.r_[np.arange(n), rng.randint(0, n, nv - n)]
                Z = np.zeros((nv, n)); Z[np.arange(nv), who] = 1.0
                yv = rng.standard_normal(nv) + 0.7 * kept[0][who]
                Ki = orc.calc_ibs_kinship(kept)
                r1 = lm.emmax(kept, list(yv), Ki, Z=Z, ctx=ctx)
                r0 = orc.emmax(kept, yv, Ki, Z=Z)
                ok0 = r0["ps"] > 1e-290
                note("emmax(Z) p", rel(r1["ps"][ok0], r0["ps"][ok0]), 1e-6, what + " values=%d" % nv)
        except Exception as e:                                     # noqa: report and continue
            fails += 1
            print("  EXCEPTION %s: %s: %s" % (what, type(e).__name__, str(e)[:300]), flush=True)
            traceback.print_exc(limit=4)
finally:
    shutil.rmtree(tmp, ignore_errors=True)
print("worst over %d cases: %s" % (cases, {k: "%.2e" % v for k, v in sorted(worst.items())}))
print("failures: %d" % fails)
sys.exit(1 if fails else 0)

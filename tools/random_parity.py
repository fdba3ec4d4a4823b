#!/usr/bin/env python3
"""Randomised parity sweep: small random shapes (individuals, SNPs, genotype alphabets, cofactors) through the device path
against the float64 oracle -- kinship (IBS / GRM / diploid IBS), emmax(), linear_model(), emmax_multi() -- to shake out
edge cases the fixed test shapes do not hit.  Checker only (oracle/ is test infrastructure).
    python tools/random_parity.py [cases] [seed]"""
import os, sys, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mixmogam_amd import _lib, kinship, linear_models as lm
from oracle import emmax_oracle as orc
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
if os.environ.get("MMG_PARITY_HOST"):                          # the host mirror alone (tests/fake_ctx.py: numpy stand-ins of the C ABI)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from fake_ctx import FakeContext
    ctx = FakeContext()
else:
    ctx = _lib.get_context()
rng = np.random.RandomState(seed)
worst = {}
fails = 0


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))) if a.size else 0.0


def note(key, val, tol, what):
    global fails
    worst[key] = max(worst.get(key, 0.0), val)
    if not val <= tol:
        fails += 1
        print("  FAIL %-18s %.3e > %.1e   %s" % (key, val, tol, what), flush=True)


for c in range(cases):
    n = int(rng.choice([rng.randint(3, 40), rng.randint(40, 300), rng.randint(256, 700)]))
    m = int(rng.choice([rng.randint(1, 30), rng.randint(30, 3000)]))
    alphabet = rng.choice(["binary", "diploid", "signed"])
    lo, hi = {"binary": (0, 2), "diploid": (0, 3), "signed": (-1, 2)}[alphabet]
    freq = rng.uniform(0.05, 0.95, m)
    if alphabet == "binary":
        snps = (rng.random_sample((m, n)) < freq[:, None]).astype(np.int8)
    else:
        snps = rng.randint(lo, hi, size=(m, n)).astype(np.int8)
    snps = snps[snps.std(1) > 0]
    if len(snps) == 0:
        continue
    m = len(snps)
    q = int(rng.choice([0, 0, 1, 2]))
    cof = [list(rng.standard_normal(n)) for _ in range(q)] or None
    y = rng.standard_normal(n) + (snps[rng.randint(m)] if rng.rand() < 0.7 else 0.0)
    what = "case %d: n=%d m=%d %s q=%d" % (c, n, m, alphabet, q)
    try:
        if alphabet == "binary":
            K = kinship.calc_ibs_kinship(snps, ctx=ctx)
            note("ibs K", float(np.max(np.abs(K - orc.calc_ibs_kinship(snps)))), 1e-12, what)
        elif alphabet == "diploid":
            K = kinship.calc_ibs_kinship(snps, snps_data_format='diploid_int', ctx=ctx)
            Ko = orc.scale_k(orc.ibs_diploid_unscaled(snps))
            note("diploid ibs K", float(np.max(np.abs(K - Ko))), 1e-10, what)
        else:
            K = None
        Kg = kinship.calc_ibd_kinship(snps, ctx=ctx)
        Kgo = orc.calc_ibd_kinship(snps)
        note("grm K", float(np.max(np.abs(Kg - Kgo)) / np.max(np.abs(Kgo))), 2e-9, what)
        if K is None:
            K = Kgo
        if n - (1 + q) - 1 < 2:
            continue
        res = lm.emmax(snps, list(y), K, cofactors=cof, ctx=ctx)
        ref = orc.emmax(snps, y, K, cofactors=cof)
        ok = ref["ps"] > 1e-290
        note("emmax p", rel(res["ps"][ok], ref["ps"][ok]), 1e-6, what + " delta %.3e" % (1 / ref["pseudo_heritability"] - 1 if ref["pseudo_heritability"] > 0 else np.inf))
        note("emmax h2", abs(res["pseudo_heritability"] - ref["pseudo_heritability"]), 1e-6, what)
        lr = lm.linear_model(snps, list(y), cofactors=cof, ctx=ctx)
        lo_ = orc.linear_model(snps, y, cofactors=cof)
        ok = lo_["ps"] > 1e-290
        note("linear_model p", rel(lr["ps"][ok], lo_["ps"][ok]), 1e-6, what)
        if rng.rand() < 0.4 and n <= 400:
            ys = np.vstack([y, rng.standard_normal(n), rng.standard_normal(n) + snps[0]])
            mr = lm.emmax_multi(snps, ys, K, cofactors=cof, ctx=ctx)
            mo = orc.emmax_multi(snps, ys, K, cofactors=cof)
            po = np.asarray([r_["ps"] for r_ in mo]) if isinstance(mo, list) else np.asarray(mo["ps"])
            ok = po > 1e-290
            note("emmax_multi p", rel(np.asarray(mr["ps"])[ok], po[ok]), 1e-6, what)
    except Exception as e:                                     # noqa: report and continue -- that IS the finding
        fails += 1
        print("  EXCEPTION %s: %s: %s" % (what, type(e).__name__, str(e)[:200]), flush=True)
        traceback.print_exc(limit=3)
print("worst over %d cases: %s" % (cases, {k: "%.2e" % v for k, v in worst.items()}))
print("failures: %d" % fails)
sys.exit(1 if fails else 0)

"""Free HBM before / after repeated use of the device paths of rounds 3-4 (band REML, fused GRM, device scale_k, FP4 kinship;
the kinships converted on the device, the scan's exact tier, a REML workspace fed from a kinship accumulator):
python tools/leak_check.py   -- a growing difference would be a leak."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixmogam_amd import _lib, linear_models as lm
hip = C.CDLL("libamdhip64.so")
def free_bytes():
    f, t = C.c_size_t(0), C.c_size_t(0)
    assert hip.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
    return f.value
ctx = _lib.get_context()
rng = np.random.RandomState(0)
n, m = 1500, 70000
snps = (rng.random_sample((m, n)) < rng.uniform(0.1, 0.9, size=(m, 1))).astype(np.int8)
snps = snps[snps.std(1) > 0]
B = rng.standard_normal((n, 64)); K = B @ B.T / 64 + 0.5 * np.eye(n)
y = rng.standard_normal(n); X = np.ones((n, 1))
few = snps[:, rng.randint(0, 12, n)][:, :]                       # every individual one of 12 genotype vectors
few = np.ascontiguousarray(snps[:4000, :12][:, rng.randint(0, 12, n)])
few = few[few.std(1) > 0]
from mixmogam_amd import kinship as _kin
Kfew = _kin.calc_ibs_kinship(few, ctx=ctx)
def once():
    g = ctx.geno(snps)
    acc = ctx.kinship_accumulator(n); acc.add_grm(g); acc.scale_k(); acc.fetch(); acc.close()
    ctx.kinship_ibs_counts(g); ctx.kinship_indicator_counts(g, 1)
    g.close()
    r = ctx.reml(K, X, y); r.sums(np.exp(np.linspace(-5, 5, 9))); r.sums([1.0], route="chol"); r.scan_model(1.0); r.close()
    # round 4
    g = ctx.geno(snps)
    ctx.kinship_ibs(g); ctx.kinship_ibs_diploid(g)
    acc = ctx.kinship_accumulator(n); acc.add_grm(g); acc.add_grm(g); acc.scale_k()
    dk = _lib.DeviceKinship(acc, scaled=True)
    r = ctx.reml(dk, X, y); r.sums(np.array([0.5, 2.0])); r.close(); dk.host(); dk.close()
    g.close()
    res = lm.emmax(few[:3000], list(y), Kfew, ctx=ctx)           # a kinship of 12 genotype classes: the scan's exact tier
    assert ctx.scan_last_stats()["n_exact"] > 0
once()
f0 = free_bytes()
for i in range(1, 16):
    once()
    if i % 5 == 0:
        print("after %2d rounds: free HBM changed by %+.1f MB" % (i, (free_bytes() - f0) / 1e6), flush=True)

"""Where the pipelined packed ingest spends its time (tools/ingest_pipe.py [chunk]): host-side stamps around the chunk
loop of hdf5_data._resident_chunks -- wait for the prefetched store, scan, fetch."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mixmogam_amd import _lib, hdf5_data
from mixmogam_amd._lib import pack_genotypes
N, M = 5000, 1000000
csz = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
ctx = _lib.get_context()
g = ctx.geno(M=M, N=N).fill_hash(20240)
# the model of bench.py's workload (IBS kinship of the same genotypes, REML): the adaptive digit schedule applies
from mixmogam_amd import kinship, linear_models as lm
rng = np.random.RandomState(20241)
y = rng.standard_normal(N)
K = kinship.scale_k(ctx.kinship_ibs_counts(g).astype(np.float64) / (2.0 * M) + 0.5)
lmm = lm.LinearMixedModel(y, ctx=ctx)
lmm.add_random_effect(K)
est = lmm.get_estimates(lmm._get_eigen_L_(), method="REML")
prep = lmm.scan_prepare(est["H_sqrt_inv"])
ctx.scan_set_model(prep["A"], prep["w"], 0)
H0, DF = prep["h0_rss"], prep["n_p"]
packed = pack_genotypes(g.download(), 1)
src = {"c": {"raw_snps_packed": packed, "packed_bits": np.array(1), "num_indivs": np.array(N), "freqs": np.full(M, 0.5),
             "positions": np.arange(M)}}
plan = hdf5_data._chunk_plan(src, 0.1, csz)
pin = [ctx.pinned_empty(csz) for _ in range(3)]
orig = _lib.Geno.upload_packed
stamps = []
def timed_upload(self, *a, **k):
    t0 = time.perf_counter(); r = orig(self, *a, **k); stamps.append(("upload", t0, time.perf_counter())); return r
_lib.Geno.upload_packed = timed_upload
for rep in range(3):
    stamps.clear()
    T0 = time.perf_counter()
    it = hdf5_data._resident_chunks(ctx, src, plan, reuse=True)
    while True:
        t0 = time.perf_counter()
        try:
            _ci, _c, gg = next(it)
        except StopIteration:
            break
        t1 = time.perf_counter()
        ctx.scan(gg, H0, DF, fetch=False)
        t2 = time.perf_counter()
        ctx.lib.mmg_scan_fetch(ctx.h, gg.M, *[_lib._ptr(b[:gg.M]) for b in pin])
        t3 = time.perf_counter()
        gg.close()
        stamps.append(("consume", t0, t1, t2, t3))
    total = time.perf_counter() - T0
    print("rep %d: %.1f ms total for %d chunks of %d; last scan: %s" % (rep, 1e3 * total, len(plan), csz, ctx.scan_last_stats()))
    for s in sorted(stamps, key=lambda x: x[1]):
        if s[0] == "upload":
            print("   upload  %7.2f -> %7.2f ms (%.2f)" % (1e3 * (s[1] - T0), 1e3 * (s[2] - T0), 1e3 * (s[2] - s[1])))
        else:
            print("   consume wait %7.2f -> %7.2f (%.2f) scan -> %7.2f (%.2f) fetch -> %7.2f (%.2f)" % (
                1e3 * (s[1] - T0), 1e3 * (s[2] - T0), 1e3 * (s[2] - s[1]), 1e3 * (s[3] - T0), 1e3 * (s[3] - s[2]),
                1e3 * (s[4] - T0), 1e3 * (s[4] - s[3])))

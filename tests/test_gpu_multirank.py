"""GPU test (-m gpu) of the multi-rank path over REAL RCCL: one process per GPU (world = 2 when the box has two
devices, else world = 1 -- the same code path through ncclCommInitRank / ncclAllReduce / ncclAllGather with one rank),
SNP blocks sharded by mixmogam_amd.dist.shard_range.  Every rank checks that the sharded results equal the
single-rank results it computes on its own GPU from the full data, bit for bit where the arithmetic is exact
(kinship counts, scan p-values: each SNP is scanned by exactly one rank with the same model; permutation minima:
max is order independent) and to 1e-12 for the fp64 GRM sum."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
from mixmogam_amd import _lib, dist as mdist, hdf5_data, kinship, linear_models as lm
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
ctx = _lib.Context(int(os.environ["LOCAL_RANK"]))
coll = mdist.RcclCollectives(ctx, rank, world, mdist.file_bootstrap(rank, world))
assert coll.info() == (rank, world, world), coll.info()
rng = np.random.RandomState(0)
n, m = 300, 2501
pops = rng.randint(0, 3, size=n)
snps = (rng.random_sample((m, n)) < rng.uniform(0.1, 0.9, size=(m, 3))[:, pops]).astype(np.int8)
y = rng.randn(n) + snps[3] - snps[77]
m0, m1 = mdist.shard_range(m, rank, world)
g_all, g_mine = ctx.geno(snps), ctx.geno(snps[m0:m1])
# kinship: partial exact counts summed in HBM over RCCL == counts of the full data
c_all = ctx.kinship_ibs_counts(g_all)
c_sh = ctx.kinship_ibs_counts(g_mine, comm=coll.device_comm)
assert np.array_equal(c_all, c_sh)
# GRM accumulator: device-resident fp64 SUM
acc = ctx.kinship_accumulator(n)
mean, sd = g_mine.snp_stats()
keep = sd > 0
gk = ctx.geno(snps[m0:m1][keep])
acc.add(gk, 1.0 / sd[keep], -mean[keep] / sd[keep])
acc.allreduce(coll.device_comm)
k_sh, cnt = acc.fetch()
mean_a, sd_a = g_all.snp_stats()
ka = sd_a > 0
g_ka = ctx.geno(snps[ka])
k_all = ctx.kinship_affine(g_ka, 1.0 / sd_a[ka], -mean_a[ka] / sd_a[ka])
assert cnt == int(ka.sum()) and np.max(np.abs(k_sh - k_all)) < 1e-9 * np.max(np.abs(k_all))
# replicas: eigh + REML; scan: SNP shards + RCCL all-gather == the scan of everything
K = kinship.scale_k(c_all / (2.0 * m) + 0.5)
lmm = lm.LinearMixedModel(y, ctx=ctx)
lmm.add_random_effect(K)
est = lmm.get_estimates(lmm._get_eigen_L_(), method="REML")
prep = lmm.scan_prepare(est["H_sqrt_inv"])
ctx.scan_set_model(prep["A"], prep["w"], 4)
full = ctx.scan(g_all, prep["h0_rss"], prep["n_p"])
count = max(b - a for a, b in (mdist.shard_range(m, r, world) for r in range(world)))
g_pad = ctx.geno(np.vstack([snps[m0:m1], np.zeros((count - (m1 - m0), n), dtype=np.int8)]))
ctx.scan(g_pad, prep["h0_rss"], prep["n_p"], fetch=False)
rss, F, p = coll.allgather_scan(count)
assert np.array_equal(mdist.unpad_gathered(p, m, world), full["ps"])
assert np.array_equal(mdist.unpad_gathered(rss, m, world), full["rss"])
# permutations: MAX all-reduce of the P statistics in HBM == the test over all SNPs
idx = np.array([np.random.RandomState(5 + q).permutation(n) for q in range(70)])
H = np.asarray(est["H_sqrt_inv"])
r0 = prep["r"]
Ys = np.ascontiguousarray(r0[idx].T)
mn_all = ctx.perm(g_all, H, Ys, prep["h0_rss"])
mn_sh = ctx.perm(g_mine, H, Ys, prep["h0_rss"], comm=coll.device_comm)
assert np.array_equal(mn_all, mn_sh)
# multi-phenotype: SNP shards + all-gather of the [P x M] blocks == unsharded
ys = rng.randn(5, n) + snps[[3, 9, 27, 81, 243]]
one = lm.emmax_multi(snps, ys, K, ctx=ctx)
two = lm.emmax_multi(snps, ys, K, ctx=ctx, coll=coll)
assert np.array_equal(one["ps"], two["ps"])
# chunked driver with chunks dealt round-robin, owned blocks gathered
src = {"c%%d" %% c: {"raw_snps": snps[c * 834:(c + 1) * 834], "freqs": snps[c * 834:(c + 1) * 834].mean(1),
                   "positions": np.arange(len(snps[c * 834:(c + 1) * 834]))} for c in range(3)}
solo = hdf5_data.run_emmax(src, y, min_maf=0.05, chunk_size=200, ctx=ctx)
both = hdf5_data.run_emmax(src, y, min_maf=0.05, chunk_size=200, ctx=ctx, coll=coll, k=solo["kinship"])
for c in solo["chrom_results"]:
    assert np.array_equal(both["chrom_results"][c]["ps"], hdf5_data.run_emmax(
        src, y, min_maf=0.05, chunk_size=200, ctx=ctx, k=solo["kinship"])["chrom_results"][c]["ps"])
bk = hdf5_data.run_emmax(src, y, min_maf=0.05, chunk_size=200, ctx=ctx, coll=coll)
assert np.max(np.abs(bk["kinship"] - solo["kinship"])) < 1e-9
# streamed multi-phenotype driver: chunks round-robin, owned [P x rows] blocks gathered per phenotype
m_solo = hdf5_data.run_emmax_multi(src, None, phenotypes=ys[:3], min_maf=0.05, chunk_size=200, ctx=ctx, k=solo["kinship"])
m_both = hdf5_data.run_emmax_multi(src, None, phenotypes=ys[:3], min_maf=0.05, chunk_size=200, ctx=ctx, k=solo["kinship"],
                                   coll=coll)
for c in m_solo["chrom_results"]:
    assert np.array_equal(m_both["chrom_results"][c]["ps"], m_solo["chrom_results"][c]["ps"])
# eigendecomposition-free REML: the grid values of delta dealt out to the ranks, sums all-gathered over RCCL
lmm_a = lm.LinearMixedModel(y, ctx=ctx); lmm_a.add_random_effect(K)
lmm_b = lm.LinearMixedModel(y, ctx=ctx); lmm_b.add_random_effect(K)
ra, rb = lmm_a.get_estimates_eigen_free(coll=coll), lmm_b.get_estimates_eigen_free()
assert ra["delta"] == rb["delta"] and ra["max_ll"] == rb["max_ll"], (ra["delta"], rb["delta"])
ra.pop("reml").close(); rb.pop("reml").close()
coll.barrier()
coll.close()
print("rank", rank, "of", world, "ok")
'''


def test_sharded_paths_over_rccl(tmp_path):
    from mixmogam_amd import _lib
    import ctypes as C
    lib = _lib.load()
    n = C.c_int(0)
    assert lib.mmg_device_count(C.byref(n)) == 0 and n.value >= 1
    world = 2 if n.value >= 2 else 1
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", WORLD_SIZE=str(world),
               MMG_RUN_ID="t%d" % os.getpid(), HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o[-3000:]
        assert "rank %d of %d ok" % (r, world) in o

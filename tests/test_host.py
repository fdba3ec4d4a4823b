"""CPU-side tests (-m "not gpu"): the C-ABI library builds, loads and exports every symbol the
header declares; the host package fails loudly without a GPU; sharding logic; simulations."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()
    from mixmogam_amd import _lib
    return _lib


def test_header_symbols_exported_and_bound(built):
    lib = built.load()
    header = open(built.HEADER_PATH).read()
    declared = set(re.findall(r"^\s*(?:int|const char\*)\s+(mmg_[a-z0-9_]+)\s*\(", header, flags=re.M))
    assert len(declared) >= 30
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert declared == set(built.PROTOTYPES), declared ^ set(built.PROTOTYPES)
    out = subprocess.run(["nm", "-D", "--defined-only", built.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" [TW] ([A-Za-z_][A-Za-z0-9_]*)$", out, flags=re.M))
    # built with -fvisibility=hidden: the header IS the export list, nothing else leaves the shared object (round 4 leaked
    # mmg_reml_create_dev); the only other dynamic symbols are the loader's own
    assert declared <= exported, declared - exported
    # (C++-mangled weak symbols are libstdc++ template instantiations, whose namespace carries default visibility itself)
    extra = {s for s in exported - declared if not s.startswith(("_Z", "_init", "_fini", "__hip", "_edata", "_end", "__bss_start"))}
    assert not extra, sorted(extra)[:20]
    assert lib.mmg_version() >= 100
    # ... and INTEGRATION.md section 9 maps every exported symbol to the reference site it replaces
    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    mapped = set(re.findall(r"^\| `(mmg_[a-z0-9_]+)` \|", integ, flags=re.M))
    assert declared <= mapped, sorted(declared - mapped)
    assert mapped <= declared, sorted(mapped - declared)


def test_code_object_targets_gfx950(built, tmp_path):
    import shutil
    copy = str(tmp_path / "lib.so")          # --offloading drops the extracted code objects beside its input
    shutil.copy(built.LIB_PATH, copy)
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", copy],
                         capture_output=True, text=True, cwd=str(tmp_path))
    txt = out.stdout + out.stderr
    if "gfx" in txt:
        assert "gfx950" in txt


def test_no_silent_cpu_fallback(built):
    import ctypes as C
    lib = built.load()
    n = C.c_int(-1)
    lib.mmg_device_count(C.byref(n))
    if n.value > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(built.MixmogamHipError):
        built.Context(0)
    from mixmogam_amd import kinship
    with pytest.raises(built.MixmogamHipError):
        kinship.calc_ibs_kinship(np.zeros((4, 8), dtype=np.int8))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "mixmogam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no CPU oracle", "").replace("The CPU oracle lives in oracle/", "") \
                    or f == "_lib.py", f


def test_shard_range_partitions():
    from mixmogam_amd.dist import shard_range, pad_block, unpad_gathered
    for total in (0, 1, 7, 8, 1000003):
        for world in (1, 2, 3, 8):
            edges = [shard_range(total, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == total
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in edges]
            assert max(sizes) - min(sizes) <= 1
    total, world = 11, 3
    x = np.arange(total, dtype=float)
    count = max(b - a for a, b in (shard_range(total, r, world) for r in range(world)))
    gathered = np.concatenate([pad_block(x[slice(*shard_range(total, r, world))], count) for r in range(world)])
    assert np.array_equal(unpad_gathered(gathered, total, world), x)


def test_scale_k_and_prepare_k_host():
    from mixmogam_amd import kinship
    from oracle import emmax_oracle as orc
    rng = np.random.RandomState(0)
    a = rng.rand(20, 20)
    k = a @ a.T
    assert np.allclose(kinship.scale_k(k), orc.scale_k(k), rtol=1e-14)
    acc = list("abcdefghijklmnopqrst")
    sub = kinship.prepare_k(k, acc, ["c", "a", "zz", "t"])
    assert np.array_equal(sub, k[[2, 0, 19]][:, [2, 0, 19]])


def test_simulations_restate_reference_shapes():
    from mixmogam_amd import simulations
    sd = simulations.simulate_genotypes(num_indivs=50, num_snps=200, seed=1)
    snps = sd['snps']
    assert snps.dtype == np.int8 and snps.shape[1] == 50 and set(np.unique(snps)) <= {0, 1}
    assert np.all(snps.sum(1) > 0)
    assert len(sd['positions']) == len(snps) == len(sd['chromosomes'])
    y = simulations.simulate_phenotype(snps, seed=2)
    assert abs(y.mean()) < 1e-12 and abs(y.std() - 1) < 1e-12


WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np
import torch.distributed as dist
from mixmogam_amd import dist as mdist
from oracle import emmax_oracle as orc
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
sys.path.insert(0, os.path.join(%(root)r, "tests"))
from torch_coll import TorchCollectives
coll = TorchCollectives()
rng = np.random.RandomState(0)
n, m = 60, 501
snps = (rng.random_sample((m, n)) < rng.uniform(0.1, 0.9, size=(m, 1))).astype(np.int8)
y = rng.randn(n) + snps[3]
m0, m1 = mdist.shard_range(m, rank, world)
# kinship: partial exact counts + integer all-reduce
counts = mdist.sharded_ibs_counts(orc.ibs_counts(snps[m0:m1]), coll)
assert np.array_equal(counts, orc.ibs_counts(snps))
K = orc.scale_k(counts / (2.0 * m) + 0.5)
# replicas: eigh + REML; scan: SNP shards + gather
X = np.ones((n, 1))
est = orc.get_estimates(y, X, orc.scale_k(K))
prep = orc.scan_prepare(y, X, est["H_sqrt_inv"])
local = orc.scan_closed(snps[m0:m1], prep)["ps"]
count = max(b - a for a, b in (mdist.shard_range(m, r, world) for r in range(world)))
gathered = coll.allgather_host(mdist.pad_block(local, count))
ps = mdist.unpad_gathered(gathered, m, world)
assert np.array_equal(ps, orc.scan_closed(snps, prep)["ps"])
# permutations: SNP shards + MIN all-reduce
idx = np.array([np.random.RandomState(5 + p).permutation(n) for p in range(6)])
pp = orc.perm_prepare(y, X, est["H_sqrt_inv"], idx)
mn = mdist.sharded_perm_min(orc.perm_closed(snps[m0:m1], pp)["min_rss"], coll)
assert np.allclose(mn, orc.perm_closed(snps, pp)["min_rss"], rtol=1e-12)
# the product's chunked driver (hdf5_data.run_emmax / run_emmax_perm) with chunks dealt to the two ranks,
# through the numpy stand-in context: kinship partial sums all-reduced, per-chunk p-values and permutation
# minima combined -- equal to the single-rank run
sys.path.insert(0, os.path.join(%(root)r, "tests"))
from fake_ctx import FakeContext
from mixmogam_amd import hdf5_data
src = {"chr%%d" %% c: {"raw_snps": snps[c * 167:(c + 1) * 167], "freqs": snps[c * 167:(c + 1) * 167].mean(1),
                     "positions": np.arange(167) + 1000 * c} for c in range(3)}
pidx = [np.random.RandomState(9 + p).permutation(n) for p in range(5)]
CHUNK = %(chunk)d                                                 # world 8: 6 chunks for 8 ranks -- two ranks own nothing
plan = hdf5_data._chunk_plan(src, 0.05, CHUNK)
owned = [ci for ci in range(len(plan)) if ci %% world == rank]
assert (len(plan) < world) == (%(chunk)d == 100 and world == 8)
both = hdf5_data.run_emmax(src, y, min_maf=0.05, chunk_size=CHUNK, ctx=FakeContext(), coll=coll)
solo = hdf5_data.run_emmax(src, y, min_maf=0.05, chunk_size=CHUNK, ctx=FakeContext(), coll=None)
assert both["num_snps"] == solo["num_snps"]
assert np.allclose(both["kinship"], solo["kinship"], rtol=1e-12, atol=1e-12)
assert abs(both["pseudo_heritability"] - solo["pseudo_heritability"]) < 1e-9
for c in solo["chrom_results"]:
    assert np.allclose(both["chrom_results"][c]["ps"], solo["chrom_results"][c]["ps"], rtol=1e-9)
# permutations shuffle the elements of the ROTATED residual, so they depend on the sign LAPACK gives each
# eigenvector (a 1e-16 difference in K can flip one): compare on the same kinship matrix
bothp = hdf5_data.run_emmax(src, y, min_maf=0.05, chunk_size=CHUNK, ctx=FakeContext(), coll=coll, num_perm=5,
                            perm_idx=pidx, k=solo["kinship"])
solop = hdf5_data.run_emmax(src, y, min_maf=0.05, chunk_size=CHUNK, ctx=FakeContext(), coll=None, num_perm=5,
                            perm_idx=pidx, k=solo["kinship"])
assert np.allclose(bothp["perm_min_ps"], solop["perm_min_ps"], rtol=1e-9)
assert np.allclose(bothp["perm_max_f_stats"], solop["perm_max_f_stats"], rtol=1e-9)
assert bothp["threshold_05"] == solop["threshold_05"] or np.allclose(bothp["threshold_05"], solop["threshold_05"], rtol=1e-9)
# multi-phenotype scan: SNP shards + all-gather of the [P x M] blocks == unsharded (numpy stand-in context)
from mixmogam_amd import linear_models as lm
ys = rng.randn(3, n) + snps[[3, 9, 27]]
one = lm.emmax_multi(snps, ys, K, ctx=FakeContext())
two = lm.emmax_multi(snps, ys, K, ctx=FakeContext(), coll=coll)
assert np.allclose(one["ps"], two["ps"], rtol=1e-12) and two["ps"].shape == (3, m)
# eigendecomposition-free REML: the 51 grid values of delta dealt out to the two ranks, sums all-gathered
lmm_a = lm.LinearMixedModel(y, ctx=FakeContext()); lmm_a.add_random_effect(K)
lmm_b = lm.LinearMixedModel(y, ctx=FakeContext()); lmm_b.add_random_effect(K)
ra, rb = lmm_a.get_estimates_eigen_free(coll=coll), lmm_b.get_estimates_eigen_free()
assert abs(ra["delta"] / rb["delta"] - 1) < 1e-12 and abs(ra["max_ll"] - rb["max_ll"]) < 1e-9
assert ra["n_factorisations"] < rb["n_factorisations"]            # each rank factorised about 1 / world of the grid
assert ra["n_factorisations"] <= -(-51 // world) + 20
# a factorisation that fails on ONE rank only (world 2: rank 0, which holds the smallest delta of an indefinite K; world 8:
# rank 5): every rank still enters the all-gather, then all of them fall back together (advisor r2: the others used to hang)
import warnings
from mixmogam_amd import _lib
from fake_ctx import FakeReml
class BreakingReml(FakeReml):
    def sums(self, deltas):
        if rank == %(fail_rank)d:
            raise _lib.MixmogamHipError("libmixmogam_hip error -4: K + delta I is not positive definite (dpotrf info 3)")
        return FakeReml.sums(self, deltas)
class BreakingCtx(FakeContext):
    def reml(self, K, X, y):
        return BreakingReml(self, K, X, y)
lmm_c = lm.LinearMixedModel(y, ctx=BreakingCtx()); lmm_c.add_random_effect(K)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    assert lmm_c._try_eigen_free(coll=coll) is None
coll.barrier()                                                    # nobody is stuck in a collective
coll.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def _run_gloo_workers(tmp_path, world, chunk, fail_rank, port):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "chunk": chunk, "fail_rank": fail_rank})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world),
               OMP_NUM_THREADS="2" if world <= 2 else "1")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    try:
        outs = [p.communicate(timeout=600)[0] for p in procs]
    finally:
        for p in procs:                                       # exact PIDs: a rank that died would leave the others in a collective
            if p.poll() is None:
                p.kill()
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert "rank %d ok" % r in o


def test_sharding_world_size_2_gloo(tmp_path):
    _run_gloo_workers(tmp_path, 2, 50, 0, 29571)


def test_sharding_world_size_8_gloo(tmp_path):
    """The same sharding logic at the TARGET world size (BASELINE configs[3..4]: 8 GPUs of one node), over gloo on the CPU:
    ragged SNP shards (501 SNPs over 8 ranks), a chunked driver with 6 chunks for 8 ranks (ranks 6 and 7 own nothing and
    must still take part in every collective), the REML grid dealt 8 ways, and a Cholesky failure injected on rank 5
    only (all ranks must raise together; nobody may be left inside the all-gather).  RCCL itself has only ever run with
    one rank (one GPU per lease): this is the multi-rank evidence the pool allows."""
    _run_gloo_workers(tmp_path, 8, 100, 5, 29581)


def test_config1_plumbing_phenotypes_and_coordination():
    """BASELINE configs[0] plumbing: the reference's FT10 phenotype (data fixture) + coordinate step."""
    from mixmogam_amd import phenotypeData as pd, snpsdata
    phend = pd.parse_phenotype_file(os.path.join(ROOT, "tests", "golden", "at_phenotypes_ft10_ft16.csv"))
    assert phend.get_name(5) == "FT10" and len(phend.get_values(5)) == 198
    accs = sorted(set(phend.get_ecotypes(5)))[::-1] + ["not_phenotyped"]
    rng = np.random.RandomState(0)
    snps = (rng.random_sample((50, len(accs))) < 0.5).astype(np.int8)
    snps[3] = 1                                   # monomorphic -> filtered
    sd = snpsdata.construct_snps_data_set(snps, list(range(50)), [1] * 50, accs)
    before = dict(zip(phend.get_ecotypes(5), phend.get_values(5)))
    info = sd.coordinate_w_phenotype_data(phend, 5)
    assert info["n_filtered_snps"] == 1 and sd.snps.shape == (49, 198)
    assert sd.accessions == phend.get_ecotypes(5)                 # same individuals, same order
    assert all(before[e] == v for e, v in zip(phend.get_ecotypes(5), phend.get_values(5)))
    Z = phend.get_incidence_matrix(5)
    assert Z.shape == (198, 198) and Z.sum() == 198


def test_config1_coordination_vs_the_reference_run():
    """The same step against what the REFERENCE's coordinate_w_phenotype_data produced (tests/golden/ft10_config1.npz,
    examples.py:84): 210 genotyped accessions in a shuffled order, 190 of them phenotyped, 8 phenotyped ones without
    genotypes, rows that stop being binary once accessions go -- kept accessions and their order on both sides, the
    phenotype values in that order, the surviving SNPs with their positions and chromosomes."""
    from conftest import load_case
    from mixmogam_amd import phenotypeData as pd, snpsdata
    c = load_case("ft10_config1")
    phend = pd.parse_phenotype_file(os.path.join(ROOT, "tests", "golden", "at_phenotypes_ft10_ft16.csv"))
    sd = snpsdata.construct_snps_data_set(c["raw_snps"], list(c["positions"]), list(c["chromosomes"]), list(c["accessions"]))
    info = sd.coordinate_w_phenotype_data(phend, int(c["phenotype_id"]))
    assert sd.accessions == [str(a) for a in c["coord_accessions"]]
    assert [str(e) for e in phend.get_ecotypes(5)] == [str(e) for e in c["coord_ecotypes"]]
    assert np.array_equal(np.asarray(phend.get_values(5)), c["coord_values"])
    assert info["n_filtered_snps"] == len(c["raw_snps"]) - len(c["snps"]) > 0
    assert np.array_equal(np.asarray(sd.get_snps()), c["snps"])
    assert np.array_equal(np.asarray(sd.get_positions()), c["coord_positions"])
    assert np.array_equal(np.asarray(sd.get_chr_list()), c["coord_chromosomes"])


def test_bench_self_launch_dry(built):
    """`python bench.py --gpus 2` without a launcher environment starts two ranks itself: the parent
    never touches the GPU; --dry-launch makes the ranks report their rendezvous environment and exit."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MMG_RUN_ID")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"],
                         capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stderr
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["dry_launch"] and rec["world"] == 2 and len(rec["ranks"]) == 2
    by_rank = {r["RANK"]: r for r in rec["ranks"]}
    assert set(by_rank) == {"0", "1"}
    for r, d in by_rank.items():
        assert d["LOCAL_RANK"] == r and d["WORLD_SIZE"] == "2" and d["MASTER_ADDR"] == "127.0.0.1"
    assert by_rank["0"]["MASTER_PORT"] == by_rank["1"]["MASTER_PORT"] and int(by_rank["0"]["MASTER_PORT"]) > 0
    assert by_rank["0"]["MMG_RUN_ID"] == by_rank["1"]["MMG_RUN_ID"] != None   # noqa: E711
    # under a launcher (WORLD_SIZE set) nothing is spawned: the process IS the rank
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"],
                         capture_output=True, text=True, timeout=120,
                         env=dict(env, RANK="1", LOCAL_RANK="1", WORLD_SIZE="2", MASTER_PORT="1234"))
    assert out.returncode == 0 and json.loads(out.stdout)["RANK"] == "1"


def test_bench_self_launch_dry_world8_and_first_contact_checks(built, monkeypatch):
    """The 8-rank launch the driver's scaling run is (dry: nobody touches a GPU), and the checks every rank runs before anything
    is timed: device identities all-gathered and asserted distinct (two ranks on one GPU -> every rank exits non-zero), the
    RCCL probe's record, and the keys every mode's ONE JSON line carries."""
    import importlib
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MMG_RUN_ID")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--mode", "strong", "--dry-launch"],
                         capture_output=True, text=True, env=env, timeout=180)
    assert out.returncode == 0, out.stderr
    rec = json.loads(out.stdout.strip().splitlines()[-1])
    assert rec["world"] == 8 and sorted(int(r["RANK"]) for r in rec["ranks"]) == list(range(8))
    assert all(r["LOCAL_RANK"] == r["RANK"] and r["WORLD_SIZE"] == "8" and r["MASTER_ADDR"] == "127.0.0.1" for r in rec["ranks"])
    assert len({r["MASTER_PORT"] for r in rec["ranks"]}) == 1 and len({r["MMG_RUN_ID"] for r in rec["ranks"]}) == 1
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    bench = importlib.import_module("bench")

    class Coll(object):                                     # 8 ranks' worth of an all-gather, seen from rank 3
        rank, world = 3, 8

        def __init__(self, buses):
            self.buses = buses

        def allgather(self, mine):
            if len(mine) != 3:                              # the probe's (rss, F, p) block
                return np.tile(mine, 8)
            rows = []
            for r in range(8):
                dom, b, rest = self.buses[r].split(":")
                dev, fn = rest.split(".")
                rows.append([r, r, (int(dom, 16) << 24) | (int(b, 16) << 16) | (int(dev, 16) << 8) | int(fn, 16)])
            assert list(mine[:2]) == [3.0, 3.0] and mine[2] == rows[3][2]
            return np.asarray(rows, dtype=np.float64).reshape(-1)

        def barrier(self):
            pass

    buses = ["0000:%02x:00.0" % (0x05 + 0x10 * r) for r in range(8)]
    ident = bench._check_distinct_devices(Coll(buses), 3, 8, 3, {"pci_bus_id": buses[3]})
    assert [i["rank"] for i in ident] == list(range(8)) and ident[3]["pci_bus_id"] == buses[3]
    assert len({i["pci_code"] for i in ident}) == 8
    dup = list(buses)
    dup[6] = dup[3]
    with pytest.raises(SystemExit) as e:
        bench._check_distinct_devices(Coll(dup), 3, 8, 3, {"pci_bus_id": dup[3]})
    assert "same GPU" in str(e.value)

    class Acc(object):
        def allreduce(self, comm):
            pass

        def close(self):
            pass

    class Ctx(object):
        def kinship_accumulator(self, n):
            return Acc()
    probe = bench.rccl_probe(Ctx(), Coll(buses), None, 5000, 125000)
    assert probe["nranks"] == 8 and probe["allreduce_bytes"] == 8.0 * 5000 * 5000
    assert probe["allreduce_busbw_gbps"] == pytest.approx(probe["allreduce_algbw_gbps"] * 2 * 7 / 8)
    for k in ("allreduce_ms", "allreduce_first_call_ms", "allgather_ms", "allgather_algbw_gbps"):
        assert probe[k] >= 0
    # every mode's line carries these (main() / bench_c5 put them there; grep the source so a rename cannot drop one silently)
    src = open(os.path.join(ROOT, "bench.py")).read()
    for key in ('"rccl_allreduce_gbps"', '"rccl_nranks"', '"end_to_end_s"', '"end_to_end_phases_s"', '"setup_s_per_rank"',
                '"devices"'):
        assert src.count(key) >= 2, key


def test_bench_self_launch_fails_cleanly_without_devices(built):
    """On a box with fewer devices than ranks the children say so and the parent exits non-zero -- no hang in the
    RCCL rendezvous, no partial JSON line."""
    if built.device_count() >= 2:
        pytest.skip("two devices present")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MMG_RUN_ID")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode != 0
    assert "needs 2 devices" in out.stderr
    assert not out.stdout.strip()


def test_committed_traffic_profile_matches_the_default_bench_arguments(monkeypatch):
    """bench.py fills roofline.traffic (scan GEMM), roofline_kinship.*.traffic and multi_phenotype.roofline.traffic from
    profiles/traffic_c3.json when that file was taken on the workload being run.  Rounds 2 and 3 both shipped `null`
    there because the file said "digits": 4 and the default run passes --digits 0: load the COMMITTED file with the
    DEFAULT arguments and insist on numbers."""
    import importlib, json, sys
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    bench = importlib.import_module("bench")
    args = bench.parse()
    t = bench._profiled_traffic(args.n, args.m, args.digits, True)
    for kernel in (bench.QUAD_KERNEL, "kinship_f32_kernel", "kinship_f4_tr_kernel", "scan_multi_mfma_kernel", "rot_gemm_w4_kernel"):
        assert t.get(kernel) and t[kernel] > 1e6, (kernel, t.get(kernel))
    assert 2e10 < t[bench.QUAD_KERNEL] < 2e11                 # tens of GB per 1M-SNP launch (11.8x the algorithmic 5.2 GB in r3)
    assert bench._profiled_traffic(args.n, args.m, 4, True) == t
    assert bench._profiled_traffic(args.n + 1, args.m, 0, True) == {}          # another workload: no borrowed numbers

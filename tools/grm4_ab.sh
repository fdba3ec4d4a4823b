#!/bin/bash
# Run on the GPU box (via gpurun): the one-pass four-plane GRM kernel (gemm_i8_grm4.h) at N=5000 x M=1e6 under different tile
# orders (MMG_GRM4_PATCH = rows x cols of 128-tiles per patch of consecutive jobs): kernel time without a profiler, then L2
# hits / misses, bytes from beyond L2 and matrix-pipe busy cycles from separate rocprofv3 --pmc passes.
#   tools/grm4_ab.sh [N] [M]   -> gpurun_out/grm4_ab/summary.txt
set -u
N=${1:-5000}; M=${2:-1000000}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/grm4_ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SUM=$OUT/summary.txt
: > $SUM
for patch in 8x16 4x8 8x8 16x16 2x2 1x32 32x1; do
  export MMG_GRM4_PATCH=$patch
  echo "== MMG_GRM4_PATCH=$patch" >> $SUM
  python3 $ROOT/tools/grm_time.py $N $M 2>&1 | tail -2 >> $SUM
  for set in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    name=$(echo $set | tr ' ' '_' | cut -c1-30)
    rm -rf $OUT/pmc_${patch}_$name
    timeout 300 rocprofv3 --pmc $set --kernel-include-regex "kinship_grm4" -d $OUT/pmc_${patch}_$name --output-format csv -- python3 $ROOT/tools/grm_time.py $N $M > $OUT/pmc_${patch}_$name.log 2>&1
  done
  python3 - "$OUT" "$patch" >> $SUM <<'PY'
import collections, csv, glob, sys
root, patch = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob("%s/pmc_%s_*/**/*counter_collection.csv" % (root, patch), recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in agg.items()}
if m:
    hit, miss = m.get("TCC_HIT_sum", 0), m.get("TCC_MISS_sum", 0)
    print("   L2 hit rate %.1f %% (hits %.3e misses %.3e of 128 B)   FETCH_SIZE x2 = %.1f GB from beyond L2   MFMA busy %.1f %% of GRBM_GUI_ACTIVE / 8"
          % (100 * hit / max(hit + miss, 1), hit, miss, 2 * m.get("FETCH_SIZE", 0) * 1024 / 1e9,
             100 * (m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024.0) / max(m.get("GRBM_GUI_ACTIVE", 1) / 8.0, 1)))
PY
done
cat $SUM

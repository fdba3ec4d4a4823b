#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats + PMC passes of the bench command.
# Usage: tools/gpu_profile.sh <tag> [bench args...]
set -u
TAG=${1:-r1}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline $*"
rocprofv3 --kernel-trace --stats -d $OUT/trace --output-format csv -- python3 $ROOT/bench.py $ARGS > $OUT/trace.log 2>&1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set -d $OUT/pmc_$name --output-format csv -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_$name.log 2>&1
done
# summaries
python3 - <<PY
import csv, glob, os, collections
out = "$OUT"
lines = []
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    lines.append("== kernel stats: " + os.path.relpath(f, out))
    lines += [l.rstrip() for l in open(f)][:25]
for f in sorted(glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        k = (row.get("Kernel_Name", "?")[:60], row.get("Counter_Name", "?"))
        agg[k][0] += float(row.get("Counter_Value", 0) or 0); agg[k][1] += 1
    lines.append("== pmc (mean per dispatch): " + os.path.relpath(f, out))
    for (kn, cn), (s, n) in sorted(agg.items()):
        if any(t in kn for t in ("scan_quad", "scan_finalize", "kinship_i8", "kinship_f32", "transpose")):
            lines.append("%-62s %-34s %16.1f  (n=%d)" % (kn, cn, s / n, n))
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:120]))
PY

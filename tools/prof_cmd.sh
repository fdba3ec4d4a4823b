#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 --kernel-trace --stats of any python command of this repo, csv output.
#   tools/prof_cmd.sh <tag> tools/reml_time.py 5000     -> gpurun_out/prof_<tag>/ (kernel_stats.csv, trace csv, the log)
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS=()
for a in "$@"; do if [ -e "$ROOT/$a" ]; then ARGS+=("$ROOT/$a"); else ARGS+=("$a"); fi; done
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/trace --output-format csv -- python3 "${ARGS[@]}" > $OUT/run.log 2>&1
grep -v "simple_timer\|rocprofv3\|generateRocpd\|tool.cpp" $OUT/run.log | tail -${PROF_TAIL:-15}
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
python3 - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print("%-70s %8s %12s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for r in rows[:28]:
    print("%-70s %8s %12.1f %10.2f %6.2f" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY

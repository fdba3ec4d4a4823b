#!/bin/bash
# GPU-box helper: shader clock held under a scan-kernel variant = GRBM_GUI_ACTIVE / 8 / kernel time.
# usage: tools/clock_probe.sh <label> [ENV=VAL ...]   (env assignments select the variant)
set -u
LABEL=$1; shift
for kv in "$@"; do export "$kv"; done
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/clock_$LABEL
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-include-regex "scan_quad" -d $OUT --output-format csv -- python3 $ROOT/tools/prof_scan.py 5000 400000 4 3 > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob
rows = [r for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f))]
for r in rows[-1:]:
    ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    print("$LABEL", "kernel ms %.3f" % ms, "GHz %.3f" % (float(r["Counter_Value"]) / 8 / (ms * 1e6)))
PY

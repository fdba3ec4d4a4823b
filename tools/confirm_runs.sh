#!/bin/bash
# Repeated runs of the sweeps that faulted in round 4 (random_parity2 / 4 / 5) and of the two stresses, one process after the
# other, stopping at the first run that does not exit 0:   bash tools/confirm_runs.sh <runs per sweep> [first seed]
# -> gpurun_out/confirm_runs.log (one line per process)
cd "$(dirname "$0")/.."
n=${1:-10}; s0=${2:-1000}
out=gpurun_out/confirm_runs.log
mkdir -p gpurun_out; : > $out
one() {
  local t0=$(date +%s)
  "$@" > gpurun_out/confirm_last.txt 2>&1
  local rc=$?
  echo "exit $rc  $(( $(date +%s) - t0 )) s  $*  | $(grep -h 'failures:\|two threads\|^stress:' gpurun_out/confirm_last.txt | tr '\n' ' ')" >> $out
  if [ $rc -ne 0 ] || grep -q "Memory access fault" gpurun_out/confirm_last.txt; then cp gpurun_out/confirm_last.txt gpurun_out/confirm_failed.txt; tail -5 $out; exit 1; fi
}
for i in $(seq 1 $n); do
  one timeout -k 10 600 python tools/random_parity2.py 30 $((s0 + i))
  one timeout -k 10 600 python tools/random_parity4.py 24 $((s0 + i))
  one timeout -k 10 600 python tools/random_parity5.py 24 $((s0 + i))
  echo "round $i of $n done"
done
one timeout -k 10 600 python tools/stress_stream.py 3000 $s0
one env MMG_STRESS_PARTS=grm_keep timeout -k 10 300 python tools/stress_two_threads.py 60 257 $s0
one env MMG_STRESS_PARTS=grm,stats,scan timeout -k 10 300 python tools/stress_two_threads.py 60 199 $s0
one env MMG_STRESS_PARTS=ibs,grm MMG_STRESS_BINARY=1 timeout -k 10 300 python tools/stress_two_threads.py 60 1001 $s0
echo "all clean: $(grep -c '^exit 0' $out) processes"

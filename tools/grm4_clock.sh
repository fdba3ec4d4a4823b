#!/bin/bash
# tools/grm4_clock.sh -- shader clock and socket power while kinship_grm4_kernel runs back to back, as shipped and under the
# ablations whose gains could be power rather than cycles (MMG_GRM4_ABL = 6: repeated A operands, 1: no DMA = stale tiles).
cd "$(dirname "$0")/.."
for abl in 0 6 1; do
  echo "---- MMG_GRM4_ABL=$abl"
  MMG_LIB="$(dirname "$0")/../mixmogam_amd/lib/libmixmogam_hip_exp.so" MMG_GRM4_ABL=$abl python3 - <<'PY' &
import os, sys, time
sys.path.insert(0, os.getcwd())
from mixmogam_amd import _lib
ctx = _lib.get_context()
g = ctx.geno(M=1000000, N=5000).fill_hash(20240)
acc = ctx.kinship_accumulator(5000)
acc.add_grm(g)
t0 = time.time(); n = 0; ms = 0.0
while time.time() - t0 < 9.0:
    acc.add_grm(g); ms += ctx.kernel_ms("grm"); n += 1
print("   %d calls, kernel %.2f ms on average" % (n, ms / n), flush=True)
PY
  PID=$!
  sleep 4
  for i in 1 2 3 4 5 6; do
    /opt/rocm/bin/rocm-smi -d 0 --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' ' | sed -E 's/ +/ /g'; echo
    sleep 0.6
  done
  wait $PID
done

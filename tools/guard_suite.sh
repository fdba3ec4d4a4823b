#!/bin/bash
# The whole GPU suite on the GUARD build (see tools/guard_sweeps.sh) -> gpurun_out/guard_suite.log
set -u
cd "$(dirname "$0")/.."
export MMG_LIB="$PWD/mixmogam_amd/lib/libmixmogam_hip_guard.so"
# the library must be current with its sources (build it on the build host first: it travels with the snapshot, a rebuild
# on the box costs GPU minutes); `make -q` answers without building
make -q -C mixmogam_amd/csrc GUARD=1 || { echo "libmixmogam_hip_guard.so is missing or older than its sources: make -C mixmogam_amd/csrc GUARD=1"; exit 1; }
echo "guard library: $(stat -c %y "$MMG_LIB")"
out=gpurun_out/guard_suite.log
mkdir -p gpurun_out
python - > $out 2>&1 <<'PY'
import ctypes, os
lib = ctypes.CDLL(os.environ["MMG_LIB"])
lib.mmg_guard_selftest.restype = ctypes.c_long
print("guard self-test (2 = both overruns seen):", lib.mmg_guard_selftest())
PY
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -p no:cacheprovider >> $out 2>&1
echo "-- exit $?" >> $out
grep -n "mmg guard\|self-test\|-- exit\|passed\|failed" $out

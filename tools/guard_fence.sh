#!/bin/bash
# Any command on the GUARD build in one of its modes (guard.hip):   bash tools/guard_fence.sh <bands|fence|fence_left> <align> <log name> <command ...>
# fence: every buffer ends on the last mapped byte of a mapping of its own, freed buffers are unmapped for good -- a read past
# an end or of a freed buffer is a GPU memory fault at that access; the abort handler prints the entry points that led there.
# Build the library first on the build host (make -C mixmogam_amd/csrc GUARD=1): it travels with the snapshot.
set -u
cd "$(dirname "$0")/.."
export MMG_LIB="$PWD/mixmogam_amd/lib/libmixmogam_hip_guard.so"
make -q -C mixmogam_amd/csrc GUARD=1 || { echo "libmixmogam_hip_guard.so is missing or older than its sources: make -C mixmogam_amd/csrc GUARD=1"; exit 1; }
export MMG_GUARD_MODE=$1 MMG_GUARD_ALIGN=$2
log=gpurun_out/$3.log
shift 3
mkdir -p gpurun_out
export MMG_GUARD_DUMP=${log%.log}.table
echo "== MMG_GUARD_MODE=$MMG_GUARD_MODE MMG_GUARD_ALIGN=$MMG_GUARD_ALIGN  $*   (library: $(stat -c %y "$MMG_LIB"))" > $log
"$@" >> $log 2>&1
rc=$?
echo "-- exit $rc" >> $log
grep -n "mmg guard\|Memory access fault\|failures\|FAIL\|EXCEPTION\|-- exit\|passed\|failed\|stress:" $log | tail -40
exit $rc

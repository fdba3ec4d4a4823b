#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats of bench.py + PMC passes restricted
# to this repo's kernels (rocSOLVER's thousands of tiny dispatches make unfiltered PMC passes take
# tens of minutes).  Usage: tools/gpu_profile.sh <tag> [bench args...]
set -u
TAG=${1:-r1}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# round 3: the extras (structured data, host ingest, multi-phenotype: rot_gemm / scan_multi) stay ON so that their kernels
# are in the same trace; `tools/gpu_profile.sh r3p --mode perm` profiles the permutation GEMM
ARGS="--steps 3 --warmup 1 --no-cpu-baseline $*"
RE="scan_quad|scan_finalize|f_sf_kernel|kinship_i8|kinship_f4|kinship_grm4|kinship_f32|transpose|perm_gemm|rot_gemm|scan_multi|grm_scale_rows|grm_combine|pack_fp4|unpack_kernel|pitch_rows"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace --output-format csv -- python3 $ROOT/bench.py $ARGS > $OUT/trace.log 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_I8" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --kernel-include-regex "$RE" -d $OUT/pmc_$name --output-format csv -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_$name.log 2>&1
done
python3 $ROOT/tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt
head -12 $OUT/trace/*/*kernel_stats.csv | cut -c1-150
cat $OUT/pmc_summary.txt | grep -E "==|scan_quad|finalize|kinship|perm_gemm|rot_gemm|scan_multi" | head -80

#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats + PMC passes (separate runs, counters only) over tools/prof_r5.py,
# restricted to the kernels of rounds 4-5.   bash tools/gpu_profile_r5.sh <tag> [N] [M]   -> gpurun_out/prof_<tag>/
set -u
TAG=${1:-r5k}; N=${2:-5000}; M=${3:-400000}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
RE="cholqr_head|gram_slices|sym_skinny|nt_update_lower|rows_gemm|band_|potrf_head|wsum|tsmm|scan_exact_den|kinship_grm4|kinship_f4_tr|perm_gemm|scan_quad|grm_scale_rows|grm_combine|quantize_kernel|center_"
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/trace --output-format csv -- python3 $ROOT/tools/prof_r5.py $N $M > $OUT/trace.log 2>&1
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_ANY"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 400 rocprofv3 --pmc $set --kernel-include-regex "$RE" -d $OUT/pmc_$name --output-format csv -- python3 $ROOT/tools/prof_r5.py $N $M > $OUT/pmc_$name.log 2>&1
  echo "pmc $name: exit $?"
done
python3 $ROOT/tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt
cp $OUT/trace/*/*kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
head -40 $OUT/kernel_stats.csv | cut -c1-160
wc -l $OUT/pmc_summary.txt

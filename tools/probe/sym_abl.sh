cd /tmp && export TMPDIR=/tmp
export MMG_LIB=$GRAFT_REPO_ROOT/mixmogam_amd/lib/libmixmogam_hip_exp.so BAND_PROF_REPS=4
for a in 0 1 2; do
  mkdir -p $GRAFT_REPO_ROOT/gpurun_out/sym_abl/a$a
  MMG_SYM_ABL=$a rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/sym_abl/a$a --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/band_prof.py > $GRAFT_REPO_ROOT/gpurun_out/sym_abl/a$a/log.txt 2>&1 || exit 1
  grep "runs:" $GRAFT_REPO_ROOT/gpurun_out/sym_abl/a$a/log.txt
done

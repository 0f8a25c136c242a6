export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
for lib in "" $R/contrastboundary_amd/lib/libcbl_amd_pp3.so $R/contrastboundary_amd/lib/libcbl_amd_pp2.so; do
  echo "== lib=$lib"
  CBL_AMD_LIB=$lib bash tools/gpu_prof_any.sh ppx 60 python $R/tools/pt_layer_time.py 40960 16 64 2>&1 | grep -E "pt_target"
  CBL_AMD_LIB=$lib bash tools/gpu_prof_any.sh ppy 60 python $R/tools/pt_layer_time.py 40960 8 32 2>&1 | grep -E "pt_target"
done

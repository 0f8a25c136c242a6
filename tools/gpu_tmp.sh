export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
for lib in "" $R/contrastboundary_amd/lib/libcbl_amd_b256.so; do
  echo "== lib=$lib"
  CBL_AMD_LIB=$lib python tools/pt_layer_time.py 40960 16 64 --graph 2>/dev/null | cut -c1-110
  CBL_AMD_LIB=$lib python tools/pt_layer_time.py 40960 8 32 --graph 2>/dev/null | cut -c1-110
  CBL_AMD_LIB=$lib bash tools/gpu_prof_any.sh ppx 60 python $R/tools/pt_layer_time.py 40960 16 64 2>&1 | grep -E "pt_(w2|agg|wstats|target)"
done

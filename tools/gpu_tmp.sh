export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
for v in "" 1; do
  echo "== CBL_PT_AGGB_ONE=$v"
  if [ -n "$v" ]; then export CBL_PT_AGGB_ONE=1; fi
  bash tools/gpu_prof_any.sh ab$v 60 python $R/tools/pt_layer_time.py 40960 16 64 2>&1 | grep -E "pt_agg_kernel<64, 16, true|pt_w2_bwd_kernel<64, 16, false"
  python tools/pt_layer_time.py 40960 16 64 --graph 2>/dev/null | cut -c1-120
  python tools/pt_layer_time.py 40960 8 32 --graph 2>/dev/null | cut -c1-120
done

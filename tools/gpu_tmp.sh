export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; export PYTHONPATH=$R
for lib in "" $R/contrastboundary_amd/lib/libcbl_amd_plain.so; do
  echo "== lib=$lib"
  CBL_AMD_LIB=$lib python bench.py --no-legs --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms', round(d['ms_per_step'],4), 'nopipe', round(d['no_pipeline']['ms_per_step'],4), 'gather us', r['launch_us'], 'frac', round(r['frac'],3), 'g200k', {k:r.get('gather_200k',{}).get(k) for k in ('launch_us','frac')})
print({k:v for k,v in r['stage_ms'].items()})
print({k:(v.get('launch_us') if isinstance(v,dict) else v) for k,v in r.items() if k in ('k4_gather','mfma_kpconv')})"
done

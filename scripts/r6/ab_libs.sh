# round 6: several experiment builds of the library against the product library, same box, in turn. usage: TAGS="t1 t2" [REPS=1] bash scripts/r6/ab_libs.sh [bench args]
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6_libs; mkdir -p $O
run() { # name, lib, args
  local n=$1; shift; local l=$1; shift
  env BABELFDTD_HIP_LIB=$l timeout 600 python bench.py --no-cpu-baseline --no-shear-workload --no-next-rows --no-group --no-production-schedule "$@" > $O/$n.json 2>$O/$n.err
  python - $O/$n.json "$n" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); rk=d.get('roofline_kernels',{})
    print(sys.argv[2], round(d['value']), round(d['ms_per_step'],4), 'step frac', round(d.get('roofline_step',{}).get('frac',0),4), {k.replace('stress_normal_solid','sns').replace('stress_shear_sparse','sss').replace('velocity_','v').replace('stress_','s'):(round(v['avg_launch_ms'],4), round(v['frac'],3)) for k,v in rk.items()})
except Exception as e: print(sys.argv[2], 'failed', e)
PY
}
for rep in $(seq ${REPS:-1}); do
  run base_$rep $PWD/babelbrain_amd/libbabelfdtd_hip.so "$@"
  for T in $TAGS; do run ${T}_$rep $PWD/babelbrain_amd/libbabelfdtd_hip_$T.so "$@"; done
done

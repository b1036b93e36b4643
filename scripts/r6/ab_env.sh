# round 6: one library, an environment switch on and off in turn on the same box. usage: VAR=BFD_VELOCITY_ALL [REPS=2] bash scripts/r6/ab_env.sh [bench args]
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6_env; mkdir -p $O
run() { # name, value, args
  local n=$1; shift; local v=$1; shift
  env $VAR=$v timeout 600 python bench.py --no-cpu-baseline --no-shear-workload --no-next-rows --no-group --no-production-schedule --no-strong-c5 --no-wide-placement "$@" > $O/$n.json 2>$O/$n.err
  python - $O/$n.json "$n" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); rk=d.get('roofline_kernels',{})
    print(sys.argv[2], round(d['value']), round(d['ms_per_step'],4), 'step frac', round(d.get('roofline_step',{}).get('frac',0),4), {k.replace('stress_normal_solid','sns').replace('stress_shear_sparse','sss').replace('velocity_','v').replace('stress_','s'):(round(v['avg_launch_ms'],4), round(v['frac'],3)) for k,v in rk.items()})
except Exception as e: print(sys.argv[2], 'failed', e); print(open(sys.argv[1].replace('.json','.err')).read()[-600:])
PY
}
for rep in $(seq ${REPS:-2}); do
  run ${VAR}_0_$rep 0 "$@"
  run ${VAR}_1_$rep 1 "$@"
done

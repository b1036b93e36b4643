# round 4 experiment: BFD_RUN_ORDER=3 (z-chunks of the absorbing layer first inside every y-band, interior chunks last) against the default order 2
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_order3; mkdir -p $O
for cfg in C3 C2; do for ord in 2 3 2 3; do
  BFD_RUN_ORDER=$ord timeout 600 python bench.py --no-cpu-baseline --no-shear-workload --no-next-rows --no-group --no-production-schedule --config $cfg --size 512 512 512 > $O/${cfg}_$ord.json 2>$O/${cfg}_$ord.err
  python - $O/${cfg}_$ord.json "${cfg}_order$ord" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); rk=d.get('roofline_kernels',{})
    print(sys.argv[2], round(d['value']), round(d['ms_per_step'],4), {k.replace('stress_normal_solid','sns').replace('stress_shear_sparse','sss').replace('velocity_','v').replace('stress_','s'):round(v['avg_launch_ms'],4) for k,v in rk.items()})
except Exception as e: print(sys.argv[2], 'failed', e)
PY
done; done | tee $O/summary.txt

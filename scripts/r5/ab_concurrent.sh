# round 5: the solid-run kernels of a half-step on side streams beside the fluid kernel (BFD_CONCURRENT=1) against the serial order
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_conc; mkdir -p $O
BFD_CONCURRENT=1 timeout 900 python -m pytest tests/test_random_media_gpu.py tests/test_parity_gpu.py -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
run() { # name, env, args
  local n=$1; shift; local e=$1; shift
  env $e timeout 600 python bench.py --no-cpu-baseline --no-shear-workload --no-next-rows --no-group "$@" > $O/$n.json 2>$O/$n.err
  python - $O/$n.json "$n" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); rk=d.get('roofline_kernels',{})
    print(sys.argv[2], round(d['value']), round(d['ms_per_step'],4), d.get('windows'), {k.replace('stress_normal_solid','sns').replace('stress_shear_sparse','sss').replace('velocity_','v').replace('stress_','s'):(round(v['avg_launch_ms'],4), round(v['frac'],3)) for k,v in rk.items()})
except Exception as e: print(sys.argv[2], 'failed', e)
PY
}
for rep in 1 2 3; do
  run C2_serial_$rep BFD_CONCURRENT=0 --config C2 --size 512 512 512
  run C2_conc_$rep BFD_CONCURRENT=1 --config C2 --size 512 512 512
done
run C4_serial BFD_CONCURRENT=0 --config C4 --scaling strong
run C4_conc BFD_CONCURRENT=1 --config C4 --scaling strong

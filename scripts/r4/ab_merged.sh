# round 4: solid runs with normal and shear stresses in one kernel (BFD_SOLID_MERGED=1) against stress_solid + stress_shear_sparse
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_merged; mkdir -p $O
BFD_SOLID_MERGED=1 timeout 1200 python -m pytest tests/test_random_media_gpu.py tests/test_parity_gpu.py tests/test_slab_gpu.py tests/test_group_gpu.py -x -q > $O/pytest_merged.txt 2>&1; tail -4 $O/pytest_merged.txt
run() { # name, env, args
  local n=$1; shift; local e=$1; shift
  env $e timeout 600 python bench.py --no-cpu-baseline --no-shear-workload --no-next-rows --no-group "$@" > $O/$n.json 2>$O/$n.err
  python - $O/$n.json "$n" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); rk=d.get('roofline_kernels',{})
    print(sys.argv[2], round(d['value']), round(d['ms_per_step'],4), {k.replace('stress_normal_solid','sns').replace('stress_shear_sparse','sss').replace('velocity_','v').replace('stress_','s'):(round(v['avg_launch_ms'],4), round(v['frac'],3)) for k,v in rk.items()})
except Exception as e: print(sys.argv[2], 'failed', e)
PY
}
for rep in 1 2; do for m in 0 1; do
  run C2_merged${m}_$rep "BFD_SOLID_MERGED=$m" --config C2 --size 512 512 512
done; done
for m in ${BIG:-}; do run C5_merged$m "BFD_SOLID_MERGED=$m" --config C5 --scaling strong --steps 60 --warmup 10 --windows 1; done

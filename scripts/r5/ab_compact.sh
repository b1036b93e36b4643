# round 5, lever (b): compact solid state (BFD_COMPACT_SOLID=1, default) against the full-volume arrays (=0), same library, same box, alternating
cd $GRAFT_REPO_ROOT; O=gpurun_out/r5_compact${TAG:+_$TAG}; mkdir -p $O
if [ -z "$SKIP_TESTS" ]; then
timeout 1200 python -m pytest tests/test_random_media_gpu.py tests/test_parity_gpu.py ${MORE_TESTS} -x -q > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
fi
run() { # name, env, args
  local n=$1; shift; local e=$1; shift
  env $e timeout 600 python bench.py --no-cpu-baseline --no-shear-workload --no-next-rows --no-group "$@" > $O/$n.json 2>$O/$n.err
  python - $O/$n.json "$n" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); rk=d.get('roofline_kernels',{})
    print(sys.argv[2], round(d['value']), round(d['ms_per_step'],4), 'step frac', round(d.get('roofline_step',{}).get('frac',0),4), {k.replace('stress_normal_solid','sns').replace('stress_shear_sparse','sss').replace('velocity_','v').replace('stress_','s'):(round(v['avg_launch_ms'],4), round(v['frac'],3)) for k,v in rk.items()})
except Exception as e: print(sys.argv[2], 'failed', e)
PY
}
for rep in $(seq ${REPS:-2}); do
  run C2_full_$rep BFD_COMPACT_SOLID=0 --config C2 --size 512 512 512
  run C2_compact_$rep BFD_COMPACT_SOLID=1 --config C2 --size 512 512 512
done
if [ -n "$WITH_C4" ]; then
run C4_full BFD_COMPACT_SOLID=0 --config C4 --scaling strong
run C4_compact BFD_COMPACT_SOLID=1 --config C4 --scaling strong
fi

# round 4: the 64 x 24 fused fluid time step (variant 4) against the default: parity test first, then C1 / C3 / C2-medium at 512^3
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_fused; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -k "fused_fluid_step" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
run() { # name, env, args
  local n=$1; shift; local e=$1; shift
  env $e timeout 600 python bench.py --no-cpu-baseline --no-shear-workload --no-next-rows "$@" > $O/$n.json 2>$O/$n.err
  python - $O/$n.json "$n" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); rk=d.get('roofline_kernels',{})
    print(sys.argv[2], round(d['value']), round(d['ms_per_step'],4), d['config'].get('tiles_rank0'), {k:(round(v['avg_launch_ms'],4), round(v['frac'],3)) for k,v in rk.items()})
except Exception as e: print(sys.argv[2], 'failed', e)
PY
}
for cfg in ${CFGS:-C3 C1 C2}; do
  run ${cfg}_v0 "A=1" --config $cfg --size 512 512 512 --variant 0
  for z in ${ZRUNS:-32}; do run ${cfg}_v4_z$z "BFD_FUSED_ZRUN=$z" --config $cfg --size 512 512 512 --variant 4; done
done

# round 4 experiment: persistent fluid kernels (BFD_PERSIST=1: 8 x 128 workgroups take run after run from per-XCD counters) against one run per workgroup
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_persist; mkdir -p $O
[ -n "$SKIPTESTS" ] || { BFD_PERSIST=1 timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_random_media_gpu.py tests/test_slab_gpu.py -x -q 2>&1 | tail -2; }
for cfg in ${CFGS:-C3 C2 C1}; do for p in 0 1 0 1; do
  BFD_PERSIST=$p BFD_PERSIST_BLOCKS=${PBLOCKS:-1024} timeout 600 python bench.py --no-cpu-baseline --no-shear-workload --no-next-rows --no-group --no-production-schedule --config $cfg --size 512 512 512 > $O/${cfg}_$p.json 2>$O/${cfg}_$p.err
  python - $O/${cfg}_$p.json "${cfg}_persist$p" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); rk=d.get('roofline_kernels',{})
    print(sys.argv[2], round(d['value']), round(d['ms_per_step'],4), {k.replace('stress_normal_solid','sns').replace('stress_shear_sparse','sss').replace('velocity_','v').replace('stress_','s'):round(v['avg_launch_ms'],4) for k,v in rk.items()})
except Exception as e: print(sys.argv[2], 'failed', e)
PY
done; done | tee $O/summary_${OUT:-run}.txt

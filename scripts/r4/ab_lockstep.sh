# round 4 experiment: fused workgroups throttled to stay within D planes of their same-XCD neighbours (BFD_FUSED_LOCKSTEP=D)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_lock; mkdir -p $O
BFD_FUSED_LOCKSTEP=1 timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -k fused_fluid_step > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
for cfg in ${CFGS:-C1 C3}; do for D in ${DS:--1 0 1 2 -1}; do
  BFD_FUSED_LOCKSTEP=$D BFD_FUSED_ZRUN=${ZRUN:-32} timeout 600 python bench.py --no-cpu-baseline --no-shear-workload --no-next-rows --no-group --config $cfg --size 512 512 512 --variant 4 > $O/${cfg}_D$D.json 2>$O/${cfg}_D$D.err
  python - $O/${cfg}_D$D.json "${cfg}_D$D" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); rk=d.get('roofline_kernels',{})
    print(sys.argv[2], round(d['value']), round(d['ms_per_step'],4), {k:(round(v['avg_launch_ms'],4), round(v['frac'],3)) for k,v in rk.items()})
except Exception as e: print(sys.argv[2], 'failed', e)
PY
done; done

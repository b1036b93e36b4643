# round 4 experiment: per-array skew of the state arrays (L2 set / channel aliasing of streams at the same cell offset), placement off
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_skew; mkdir -p $O
for cfg in ${CFGS:-C1}; do for v in 0 4; do for sk in ${SKEWS:-0 5 17 0}; do
  BFD_PLACEMENT=0 BFD_SKEW_LINES=$sk timeout 600 python bench.py --no-cpu-baseline --no-shear-workload --no-next-rows --config $cfg --size 512 512 512 --variant $v > $O/${cfg}_v${v}_s$sk.json 2>$O/${cfg}_v${v}_s$sk.err
  python - $O/${cfg}_v${v}_s$sk.json "${cfg}_v${v}_skew$sk" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); rk=d.get('roofline_kernels',{})
    print(sys.argv[2], round(d['value']), round(d['ms_per_step'],4), {k:(round(v['avg_launch_ms'],4), round(v['frac'],3)) for k,v in rk.items()})
except Exception as e: print(sys.argv[2], 'failed', e)
PY
done; done; done

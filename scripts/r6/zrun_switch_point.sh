cd $GRAFT_REPO_ROOT; O=gpurun_out/r6_zrun; mkdir -p $O
for cfg in "C2 --size 320 320 384" "C3 --size 320 320 384" "C2 --size 352 352 352" "C3 --size 352 352 352" "C3 --size 320 320 320"; do
for z in 8 16; do
  n=$(echo "${cfg}_z$z" | tr ' ' '_' | tr -d '-')
  env BFD_ZRUN=$z timeout 300 python bench.py --config $cfg --steps 200 --warmup 40 --no-cpu-baseline --no-shear-workload --no-next-rows --no-group --no-production-schedule --no-strong-c5 --no-wide-placement > $O/$n.json 2>$O/$n.err
  python - $O/$n.json "$cfg ZRUN=$z" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); rk=d.get('roofline_kernels',{})
    print(sys.argv[2], round(d['value']), round(d['ms_per_step'],4), {k.replace('stress_shear_sparse','sss').replace('velocity_','v').replace('stress_','s'):round(v['avg_launch_ms'],4) for k,v in rk.items()})
except Exception as e: print(sys.argv[2], 'failed', e)
PY
done; done

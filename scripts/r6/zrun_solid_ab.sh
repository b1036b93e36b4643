# round 6: solid runs of 8 planes beside fluid runs of 16 (BFD_ZRUN_SOLID=8), shear medium at 512^3 and C4, same box in turn
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6_zsolid; mkdir -p $O
for rep in 1 2; do
for cfg in "C2 --size 512 512 512" "C4"; do
for z in 16 8; do
  n=$(echo "${cfg}_z${z}_$rep" | tr ' ' '_' | tr -d '-')
  env BFD_ZRUN_SOLID=$z timeout 400 python bench.py --config $cfg --steps 100 --warmup 20 --no-cpu-baseline --no-shear-workload --no-next-rows --no-group --no-production-schedule --no-strong-c5 --no-wide-placement > $O/$n.json 2>$O/$n.err
  python - $O/$n.json "$cfg solid runs of $z planes" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); rk=d.get('roofline_kernels',{})
    print(sys.argv[2], round(d['value']), round(d['ms_per_step'],4), {k.replace('stress_shear_sparse','sss').replace('velocity_','v').replace('stress_','s'):round(v['avg_launch_ms'],4) for k,v in rk.items()})
except Exception as e: print(sys.argv[2], 'failed', e)
PY
done; done; done

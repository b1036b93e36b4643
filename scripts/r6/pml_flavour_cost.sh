# round 6, TIMING EXPERIMENT (results of the second line of each pair are wrong): what the absorbing-layer flavour of the fluid kernels costs. Library built from the
# patch in profiles/r6/pml_flavour_cost.txt: with BFD_EXP_NO_PML_FLAVOUR set, fluid runs that touch the layer take the interior flavour.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6_pml; mkdir -p $O
for cfg in "C3" "C3 --size 256 256 256"; do
for v in 0 1; do
  n=$(echo "${cfg}_$v" | tr ' ' '_' | tr -d '-')
  if [ $v = 1 ]; then export BFD_EXP_NO_PML_FLAVOUR=1; else unset BFD_EXP_NO_PML_FLAVOUR; fi
  env BABELFDTD_HIP_LIB=$PWD/babelbrain_amd/libbabelfdtd_hip_nopml.so timeout 300 python bench.py --config $cfg --steps 200 --warmup 40 --no-cpu-baseline --no-shear-workload --no-next-rows --no-group --no-production-schedule --no-strong-c5 --no-wide-placement > $O/$n.json 2>$O/$n.err
  python - $O/$n.json "$cfg interior flavour everywhere=$v" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); rk=d.get('roofline_kernels',{})
    print(sys.argv[2], round(d['value']), round(d['ms_per_step'],4), {k.replace('velocity_','v').replace('stress_','s'):round(v['avg_launch_ms'],4) for k,v in rk.items()})
except Exception as e: print(sys.argv[2], 'failed', e); print(open(sys.argv[1].replace('.json','.err')).read()[-500:])
PY
done; done

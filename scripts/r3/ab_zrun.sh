# z-run length of the marching kernels (BFD_ZRUN: 8 / 16 (default at this size) / 32) with the arrays placed: C3 and the shear medium at 512^3, twice
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_zrun; mkdir -p $O
for i in 1 2; do for z in 16 8 32; do
  BFD_ZRUN=$z timeout 600 python bench.py --no-cpu-baseline --no-next-rows > $O/z${z}_$i.json 2>/dev/null
  python - $O/z${z}_$i.json $z <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); s=d['shear_workload']; print('zrun', sys.argv[2], 'C3', round(d['value']), round(d['ms_per_step'],4), '| shear medium', round(s['value']), round(s['ms_per_step'],4), {k:round(v['avg_launch_ms'],3) for k,v in s['roofline_kernels'].items()})
PY
done; done

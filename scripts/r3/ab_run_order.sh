# run order of the lists (BFD_RUN_ORDER: 2 = eight y-bands, default; 1 = z-chunk slowest over the plane; 0 = column order) with the arrays placed
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_order; mkdir -p $O
for i in 1 2; do for m in 2 1 0; do
  BFD_RUN_ORDER=$m timeout 600 python bench.py --no-cpu-baseline --no-next-rows > $O/m${m}_$i.json 2>/dev/null
  python - $O/m${m}_$i.json $m <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); s=d['shear_workload']; print('order', sys.argv[2], 'C3', round(d['value']), round(d['ms_per_step'],4), '| shear medium', round(s['value']), round(s['ms_per_step'],4))
PY
done; done

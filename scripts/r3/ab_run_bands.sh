# y-bands of the run order (BFD_RUN_BANDS, default 8 = one per XCD part): does a shorter way back to the z-neighbour run help the large planes of C5 / C4?
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_bands; mkdir -p $O
for c in C5 C4; do for nb in 8 16 32 64; do
  BFD_RUN_BANDS=$nb timeout 900 python bench.py --config $c --scaling strong --no-cpu-baseline > $O/${c}_b$nb.json 2>/dev/null
  python - $O/${c}_b$nb.json $c $nb <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], 'bands', sys.argv[3], round(d['value']), round(d['ms_per_step'],3), {k:round(v['avg_launch_ms'],3) for k,v in d.get('roofline_kernels',{}).items()})
PY
done; done
for nb in 8 16 32; do
  BFD_RUN_BANDS=$nb timeout 600 python bench.py --no-cpu-baseline --no-next-rows --no-shear-workload > $O/C3_b$nb.json 2>/dev/null
  python - $O/C3_b$nb.json C3 $nb <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], 'bands', sys.argv[3], round(d['value']), round(d['ms_per_step'],3))
PY
done

# what the absorbing layer costs: water 512^3 (C1) and C3 with the prescribed 12 cells against 2 cells (fewer tiles take the layer's kernel flavour)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_pml; mkdir -p $O
for i in 1 2; do for c in C1 C3; do for nd in 12 2; do
  BENCH_NDELTA=$nd timeout 600 python bench.py --config $c --size 512 512 512 --no-cpu-baseline --no-shear-workload > $O/${c}_nd${nd}_$i.json 2>/dev/null
  python - $O/${c}_nd${nd}_$i.json $c $nd <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], 'NDelta', sys.argv[3], round(d['value']), round(d['ms_per_step'],4), d['config']['tiles_rank0'], {k:round(v['avg_launch_ms'],4) for k,v in d.get('roofline_kernels',{}).items()})
PY
done; done; done

# full GPU suite, then C4 and C5 whole on one GPU with the round's final binary
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_final; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 ) 2>&1 | tee $O/gpu_suite.txt
for c in C4 C5; do
  timeout 900 python bench.py --config $c --scaling strong --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err
  python - $O/bench_$c.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(d['config']['workload'][:60], round(d['value']), round(d['ms_per_step'],3), {k:round(v['avg_launch_ms'],3) for k,v in d.get('roofline_kernels',{}).items()}, d['config'].get('array_placement','')[:100])
PY
done

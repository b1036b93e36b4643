cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_place; mkdir -p $O
summ() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], round(d['value']), round(d['ms_per_step'],4), {k:round(v['avg_launch_ms'],3) for k,v in d.get('roofline_kernels',{}).items()})
PY
}
for rep in 1 2; do
  for c in C4 C5; do
    BFD_PLACEMENT_VERBOSE=1 python bench.py --config $c --scaling strong --steps 40 --warmup 10 --no-cpu-baseline > $O/${c}_chosen_$rep.json 2> $O/${c}_chosen_$rep.err; grep placement $O/${c}_chosen_$rep.err | tr '\n' ';'; echo; summ $O/${c}_chosen_$rep.json
    BFD_PLACEMENT_TRIALS=0 python bench.py --config $c --scaling strong --steps 40 --warmup 10 --no-cpu-baseline > $O/${c}_raw_$rep.json 2>/dev/null; summ $O/${c}_raw_$rep.json
  done
done
timeout 2400 python -m pytest tests -m gpu -q -x --durations=6 -o faulthandler_timeout=1500 > $O/tests.log 2>&1; tail -12 $O/tests.log

cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_place; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x -k "not c5_1024 and not c4_h317 and not c2_full and not rayleigh_study" -o faulthandler_timeout=600 > $O/tests.log 2>&1; grep -E "passed|failed" $O/tests.log
BFD_PLACEMENT_VERBOSE=1 python scripts/placement_probe.py 2>&1 | grep "placement\|^engine" | head -48
summ() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], round(d['value']), round(d['ms_per_step'],4), {k:round(v['avg_launch_ms'],3) for k,v in d.get('roofline_kernels',{}).items()})
PY
}
for rep in 1 2 3 4; do
  python bench.py --no-cpu-baseline --no-shear-workload --steps 300 --warmup 30 > $O/c3_tuned_$rep.json 2>/dev/null; summ $O/c3_tuned_$rep.json
  BFD_PLACEMENT_TRIALS=0 python bench.py --no-cpu-baseline --no-shear-workload --steps 300 --warmup 30 > $O/c3_untuned_$rep.json 2>/dev/null; summ $O/c3_untuned_$rep.json
done

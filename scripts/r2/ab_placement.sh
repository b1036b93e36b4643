# Placement choice (bfd_prepare): the probe with its candidates printed, then the default bench against
# BFD_PLACEMENT_TRIALS=0, interleaved, C3 and the shear workload.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_place; mkdir -p $O
BFD_PLACEMENT_TRIALS=0 python scripts/placement_probe.py 2>&1 | grep "^engine" > $O/probe_raw.txt
BFD_PLACEMENT_VERBOSE=1 python scripts/placement_probe.py 2>&1 | grep "placement\|^engine" > $O/probe_chosen.txt
cat $O/probe_raw.txt; grep "kept\|^engine" $O/probe_chosen.txt
summ() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], round(d['value']), round(d['ms_per_step'],4), {k:round(v['avg_launch_ms'],3) for k,v in d.get('roofline_kernels',{}).items()})
PY
}
for rep in 1 2 3 4 5 6; do
  python bench.py --no-cpu-baseline --no-shear-workload --steps 300 --warmup 30 > $O/c3_chosen_$rep.json 2>/dev/null; summ $O/c3_chosen_$rep.json
  BFD_PLACEMENT_TRIALS=0 python bench.py --no-cpu-baseline --no-shear-workload --steps 300 --warmup 30 > $O/c3_raw_$rep.json 2>/dev/null; summ $O/c3_raw_$rep.json
done
for rep in 1 2 3; do
  python bench.py --config C2 --size 512 512 512 --no-cpu-baseline --steps 200 --warmup 30 > $O/c2_chosen_$rep.json 2>/dev/null; summ $O/c2_chosen_$rep.json
  BFD_PLACEMENT_TRIALS=0 python bench.py --config C2 --size 512 512 512 --no-cpu-baseline --steps 200 --warmup 30 > $O/c2_raw_$rep.json 2>/dev/null; summ $O/c2_raw_$rep.json
done

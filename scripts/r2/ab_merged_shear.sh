cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_run11; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x -k "not c5_1024 and not c4_h317 and not c2_full and not rayleigh_study" -o faulthandler_timeout=600 > $O/tests.log 2>&1; tail -4 $O/tests.log
summ() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], round(d['value']), round(d['ms_per_step'],4), {k:round(v['avg_launch_ms'],3) for k,v in d.get('roofline_kernels',{}).items()})
PY
}
for rep in 1 2; do
  python bench.py --config C2 --size 512 512 512 --no-cpu-baseline --steps 200 --warmup 30 > $O/c2_merged_$rep.json 2>/dev/null; summ $O/c2_merged_$rep.json
  BFD_SHEAR_SPARSE=1 python bench.py --config C2 --size 512 512 512 --no-cpu-baseline --steps 200 --warmup 30 > $O/c2_sparse_$rep.json 2>/dev/null; summ $O/c2_sparse_$rep.json
done
python bench.py --config C5 --scaling strong --steps 60 --warmup 10 --no-cpu-baseline > $O/c5_merged.json 2>/dev/null; summ $O/c5_merged.json
BFD_SHEAR_SPARSE=1 python bench.py --config C5 --scaling strong --steps 60 --warmup 10 --no-cpu-baseline > $O/c5_sparse.json 2>/dev/null; summ $O/c5_sparse.json

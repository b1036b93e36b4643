# sparse shear list order: 0 ascending cell index, 1 tile-run order, 2 (z-chunk, band of 8 rows, plane, row, i): shear medium 512^3 and C5, kernel ms
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_shear_order; mkdir -p $O
run() { local n=$1; shift; local so=$1; shift
  BFD_SHEAR_ORDER=$so timeout 900 python bench.py --no-cpu-baseline --no-next-rows "$@" > $O/$n.json 2>/dev/null
  python - $O/$n.json $n <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d['value']), round(d['ms_per_step'],4), {k:round(v['avg_launch_ms'],4) for k,v in d.get('roofline_kernels',{}).items()})
PY
}
for i in 1 2; do for so in 0 1 2; do run c2_order${so}_$i $so --config C2 --size 512 512 512; done; done
for so in 0 1 2; do run c5_order$so $so --config C5 --scaling strong; done

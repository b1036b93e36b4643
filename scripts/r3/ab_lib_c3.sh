# same-box A/B of the product library against a variant build (make TAG=<name>) on the default bench line (C3 + shear medium), alternating
cd $GRAFT_REPO_ROOT; T=$1; O=gpurun_out/r3_abc3_$T; mkdir -p $O
for i in $(seq 1 ${2:-3}); do for lib in "" $T; do
  L=""; [ -n "$lib" ] && L=$GRAFT_REPO_ROOT/babelbrain_amd/libbabelfdtd_hip_$lib.so
  BABELFDTD_HIP_LIB=$L timeout 600 python bench.py --no-cpu-baseline --no-next-rows > $O/${lib:-product}_$i.json 2>/dev/null
  python - $O/${lib:-product}_$i.json ${lib:-product} <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); s=d['shear_workload']; print(sys.argv[2], 'C3', round(d['value']), round(d['ms_per_step'],4), {k:round(v['avg_launch_ms'],4) for k,v in d.get('roofline_kernels',{}).items()}, '| shear medium', round(s['value']), round(s['ms_per_step'],4))
PY
done; done

# same-box A/B of the product library against a variant build (make TAG=<name>): shear workload 512^3, alternating; then the parity tests on the product library
# usage: bash scripts/r3/ab_lib.sh <tag> [rounds]
cd $GRAFT_REPO_ROOT; T=$1; O=gpurun_out/r3_ab_$T; mkdir -p $O
for i in $(seq 1 ${2:-3}); do for lib in "" $T; do
  L=""; [ -n "$lib" ] && L=$GRAFT_REPO_ROOT/babelbrain_amd/libbabelfdtd_hip_$lib.so
  BABELFDTD_HIP_LIB=$L timeout 600 python bench.py --config C2 --size 512 512 512 --no-cpu-baseline --no-next-rows > $O/c2_${lib:-product}_$i.json 2>/dev/null
  python - $O/c2_${lib:-product}_$i.json ${lib:-product} <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d['value']), round(d['ms_per_step'],4), {k:round(v['avg_launch_ms'],4) for k,v in d.get('roofline_kernels',{}).items()})
PY
done; done
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_random_media_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -2

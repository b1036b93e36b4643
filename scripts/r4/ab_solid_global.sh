# round 4: the solid-run kernels with GLOBAL loads and a branch-free prefetch (-DBFD_SOLID_GLOBAL build) against the shipped ones
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_sg; mkdir -p $O
L=$PWD/babelbrain_amd/libbabelfdtd_hip_${TAG:-sg}.so
BABELFDTD_HIP_LIB=$L timeout 1200 python -m pytest tests/test_random_media_gpu.py tests/test_parity_gpu.py tests/test_slab_gpu.py -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
run() { # name, lib, args
  local n=$1; shift; local l=$1; shift
  BABELFDTD_HIP_LIB=$l timeout 600 python bench.py --no-cpu-baseline --no-shear-workload --no-next-rows --no-group "$@" > $O/$n.json 2>$O/$n.err
  python - $O/$n.json "$n" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); rk=d.get('roofline_kernels',{})
    print(sys.argv[2], round(d['value']), round(d['ms_per_step'],4), {k.replace('stress_normal_solid','sns').replace('stress_shear_sparse','sss').replace('velocity_','v').replace('stress_','s'):(round(v['avg_launch_ms'],4), round(v['frac'],3)) for k,v in rk.items()})
except Exception as e: print(sys.argv[2], 'failed', e)
PY
}
for rep in 1 2; do
  run C2_base_$rep $PWD/babelbrain_amd/libbabelfdtd_hip.so --config C2 --size 512 512 512
  run C2_new_$rep $L --config C2 --size 512 512 512
done

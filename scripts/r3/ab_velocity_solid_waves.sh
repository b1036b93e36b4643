# velocity_solid (plain flavour) at 6 waves/SIMD (80 VGPRs, 6 spilled) against 4 (86 VGPRs): same box, shear workload 512^3
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_w6; mkdir -p $O
for i in 1 2 3; do for lib in "" w6; do
  L=""; [ -n "$lib" ] && L=$GRAFT_REPO_ROOT/babelbrain_amd/libbabelfdtd_hip_$lib.so
  BABELFDTD_HIP_LIB=$L timeout 600 python bench.py --config C2 --size 512 512 512 --no-cpu-baseline --no-next-rows > $O/c2_${lib:-base}_$i.json 2>/dev/null
  python - $O/c2_${lib:-base}_$i.json ${lib:-base} <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d['value']), d['ms_per_step'], {k:round(v['avg_launch_ms'],4) for k,v in d.get('roofline_kernels',{}).items()})
PY
done; done

# round 4: experiment builds of the fused kernel (libbabelfdtd_hip_<tag>.so), bench at 512^3 with variant 4: what bounds it
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_exp; mkdir -p $O
for tag in ${TAGS:-base noring occ1 plainst}; do
  lib=babelbrain_amd/libbabelfdtd_hip.so; [ $tag != base ] && lib=babelbrain_amd/libbabelfdtd_hip_$tag.so
  for cfg in ${CFGS:-C1}; do
    BABELFDTD_HIP_LIB=$PWD/$lib timeout 600 python bench.py --no-cpu-baseline --no-shear-workload --no-next-rows --config $cfg --size 512 512 512 --variant ${VARIANT:-4} ${BENCH_ARGS} > $O/${cfg}_$tag.json 2>$O/${cfg}_$tag.err
    python - $O/${cfg}_$tag.json "${cfg}_$tag" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); rk=d.get('roofline_kernels',{})
    print(sys.argv[2], round(d['value']), round(d['ms_per_step'],4), {k:(round(v['avg_launch_ms'],4), round(v['frac'],3)) for k,v in rk.items()})
except Exception as e: print(sys.argv[2], 'failed', e)
PY
  done
done

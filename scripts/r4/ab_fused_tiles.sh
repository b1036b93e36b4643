# round 4 experiment: the fused time step on smaller output tiles (64 x 8 with two cells per thread, 64 x 16 with three) against 64 x 24 / four
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_ftiles; mkdir -p $O
for tag in ${TAGS:-fs8 fs16}; do
  BABELFDTD_HIP_LIB=$PWD/babelbrain_amd/libbabelfdtd_hip_$tag.so timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -k fused_fluid_step > $O/pytest_$tag.txt 2>&1; echo "$tag: $(tail -1 $O/pytest_$tag.txt)"
done
for cfg in ${CFGS:-C1 C3}; do
  for tag in base ${TAGS:-fs8 fs16}; do for z in ${ZRUNS:-32 64}; do
    lib=babelbrain_amd/libbabelfdtd_hip.so; [ $tag != base ] && lib=babelbrain_amd/libbabelfdtd_hip_$tag.so
    BFD_FUSED_ZRUN=$z BABELFDTD_HIP_LIB=$PWD/$lib timeout 600 python bench.py --no-cpu-baseline --no-shear-workload --no-next-rows --no-group --config $cfg --size 512 512 512 --variant 4 > $O/${cfg}_${tag}_$z.json 2>$O/${cfg}_${tag}_$z.err
    python - $O/${cfg}_${tag}_$z.json "${cfg}_${tag}_z$z" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); rk=d.get('roofline_kernels',{})
    print(sys.argv[2], round(d['value']), round(d['ms_per_step'],4), d['config']['tiles_rank0']['fused_fluid'], {k:(round(v['avg_launch_ms'],4), round(v['frac'],3)) for k,v in rk.items()})
except Exception as e: print(sys.argv[2], 'failed', e)
PY
  done; done
done

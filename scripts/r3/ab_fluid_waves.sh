#!/bin/bash
# velocity_fluid<true> spills one register at the 8 waves/SIMD cap (64 VGPRs, 8 B of scratch): A/B against 7 waves/SIMD (72 VGPRs)
# build first (here or on the box): make -C babelbrain_amd/csrc TAG=w7 EXTRA=-DFLUID_WAVES_PER_SIMD=7
mkdir -p gpurun_out/r3
for i in 1 2 3; do
  for lib in "" _w7; do
    BABELFDTD_HIP_LIB=$PWD/babelbrain_amd/libbabelfdtd_hip$lib.so python bench.py --steps 200 --warmup 30 --no-cpu-baseline --no-shear-workload > gpurun_out/r3/ab_waves${lib}_$i.json 2>/dev/null
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r3/ab_waves*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1]); k = d['roofline_kernels']
    print('%-28s C3 %.1f Gvoxel-steps/s  velocity_fluid %.3f ms  stress_fluid %.3f ms' % (f.split('/')[-1], d['value'] / 1e3, k['velocity_fluid']['avg_launch_ms'], k['stress_fluid']['avg_launch_ms']))
PY

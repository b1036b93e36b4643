#!/bin/bash
# EXPERIMENT (timing only): the ten solid-only arrays tile-blocked in two allocations (build: make -C babelbrain_amd/csrc TAG=blk
# EXTRA=-DBFD_EXP_BLOCKED) against the shipping layout, shear medium 512^3, interleaved on one box
mkdir -p gpurun_out/r3
for i in 1 2 3; do
  for lib in "" _blk; do
    BABELFDTD_HIP_LIB=$PWD/babelbrain_amd/libbabelfdtd_hip$lib.so python bench.py --config C2 --size 512 512 512 --steps 150 --warmup 30 --no-cpu-baseline > gpurun_out/r3/ab_blocked${lib}_$i.json 2>/dev/null
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r3/ab_blocked*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1]); k = d['roofline_kernels']
    print('%-26s %.1f Gvoxel-steps/s  %s' % (f.split('/')[-1], d['value'] / 1e3, {c: round(v['avg_launch_ms'], 3) for c, v in k.items()}))
    if 'EXPERIMENT' in d['config']['array_placement']: print('   ', d['config']['array_placement'].split('EXPERIMENT')[1])
PY

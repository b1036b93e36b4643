#!/bin/bash
# Region-aware placement of the arrays (bfd_prepare, round 3) against BFD_PLACEMENT=0, interleaved on one box.
# usage: scripts/r3/placement_ab.sh [runs]   -> gpurun_out/r3/placement_ab_*.json / .err
mkdir -p gpurun_out/r3
n=${1:-2}
for i in $(seq 1 $n); do
  BFD_PLACEMENT_VERBOSE=1 python bench.py --steps 200 --warmup 30 --no-cpu-baseline > gpurun_out/r3/placement_ab_on_$i.json 2> gpurun_out/r3/placement_ab_on_$i.err
  BFD_PLACEMENT=0 python bench.py --steps 200 --warmup 30 --no-cpu-baseline > gpurun_out/r3/placement_ab_off_$i.json 2> gpurun_out/r3/placement_ab_off_$i.err
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r3/placement_ab_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        k = d['roofline_kernels']
        s = d.get('shear_workload', {})
        print('%-44s C3 %.1f Gvox/s (vel %.3f str %.3f ms)  shear %.1f  %s' % (f.split('/')[-1], d['value'] / 1e3, k['velocity_fluid']['avg_launch_ms'], k['stress_fluid']['avg_launch_ms'],
              (s.get('value') or 0) / 1e3, {c: round(v['avg_launch_ms'], 3) for c, v in (s.get('roofline_kernels') or {}).items()}))
    except Exception as e:
        print(f, 'unreadable', e)
PY
grep -h placement gpurun_out/r3/placement_ab_on_*.err

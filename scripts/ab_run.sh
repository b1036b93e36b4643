#!/bin/bash
# on the GPU box: alternate A and B several times (box-to-box variance is +-5 %, within a box ~1-4 %)
cfgs=${1:-"C1 C3"}; reps=${2:-2}; size=${3:-"512 512 512"}
for r in $(seq $reps); do for L in A B; do for c in $cfgs; do
BABELFDTD_HIP_LIB=$PWD/ab/lib$L.so python bench.py --config $c --size $size --steps 300 --warmup 50 --no-cpu-baseline --no-dense-reference 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L', d['config']['workload'][:3], round(d['value']), round(d['ms_per_step'],4), 'stress/vel ms', sorted([round(d['roofline']['avg_launch_ms'],4), round(d['roofline_other']['avg_launch_ms'],4)]))"
done; done; done

#!/bin/bash
# quick A/B on the GPU box: parity subset + bench lines for C1/C3 at 512^3 (and C2 if asked)
cfgs=${1:-"C1 C3"}
timeout 600 python -m pytest tests/test_parity_gpu.py -q -m gpu -x 2>&1 | tail -2
for c in $cfgs; do python bench.py --config $c --size 512 512 512 --steps 60 --warmup 6 --no-cpu-baseline --no-dense-reference 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:3], round(d['value']), round(d['ms_per_step'],4), 'stress/vel ms', sorted([round(d['roofline']['avg_launch_ms'],4), round(d['roofline_other']['avg_launch_ms'],4)]))"; done

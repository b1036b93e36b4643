# same box: default variant against variant 4 (fused fluid step)
cfgs=${1:-"C1 C3 C2"}; reps=${2:-2}
for r in $(seq $reps); do for v in 3 4; do for c in $cfgs; do
python bench.py --config $c --size 512 512 512 --variant $v --steps 60 --warmup 6 --no-cpu-baseline --no-dense-reference 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('variant $v', d['config']['workload'][:3], round(d['value']), round(d['ms_per_step'],4), 'stress/vel ms', [round(d['roofline']['avg_launch_ms'],4), round(d['roofline_other']['avg_launch_ms'],4)], d['config']['tiles_rank0'].get('fused_fluid'))"
done; done; done

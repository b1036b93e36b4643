for r in 1 2; do for z in 16 32 64 128; do for c in C1 C3; do
BFD_FUSED_ZRUN=$z python bench.py --config $c --size 512 512 512 --variant 4 --steps 60 --warmup 6 --no-cpu-baseline --no-dense-reference 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fused zrun $z', d['config']['workload'][:3], round(d['value']), round(d['ms_per_step'],4), 'stress/vel ms', [round(d['roofline']['avg_launch_ms'],4), round(d['roofline_other']['avg_launch_ms'],4)])"
done; done; done
python bench.py --config C3 --size 512 512 512 --variant 3 --steps 60 --warmup 6 --no-cpu-baseline --no-dense-reference 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('variant 3', d['config']['workload'][:3], round(d['value']), round(d['ms_per_step'],4))"

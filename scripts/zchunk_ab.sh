# effect of the run length (z-chunk) under the current run order
for r in 1 2; do for z in 32 16 8; do
  for c in C1 C3; do BFD_ZRUN=$z python bench.py --config $c --size 512 512 512 --steps 60 --warmup 6 --no-cpu-baseline --no-dense-reference 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('zrun $z', d['config']['workload'][:3], round(d['value']), round(d['ms_per_step'],4), 'stress/vel ms', sorted([round(d['roofline']['avg_launch_ms'],4), round(d['roofline_other']['avg_launch_ms'],4)]))"; done
done; done

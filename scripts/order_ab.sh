for r in 1 2; do for o in 0 1 2; do for c in C1 C3; do
BFD_RUN_ORDER=$o python bench.py --config $c --size 512 512 512 --steps 60 --warmup 6 --no-cpu-baseline --no-dense-reference 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('order $o', d['config']['workload'][:3], round(d['value']), round(d['ms_per_step'],4), 'stress/vel ms', sorted([round(d['roofline']['avg_launch_ms'],4), round(d['roofline_other']['avg_launch_ms'],4)]))"
done; done; done
for o in 1 2; do BFD_RUN_ORDER=$o bash scripts/pmc_traffic.sh c1_order$o --config C1 --size 512 512 512 | grep -E "^==|HBM"; done

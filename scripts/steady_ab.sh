# A/B at steady clocks (default K/W = 600/300), same box
run() { # label, env..., -- bench args
  label=$1; shift
  env "$@" python bench.py --no-cpu-baseline --no-dense-reference $BARGS 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$label', d['config']['workload'][:3], round(d['value']), round(d['ms_per_step'],4), 'stress/vel ms', [round(d['roofline']['avg_launch_ms'],4), round(d['roofline_other']['avg_launch_ms'],4)])"
}
for r in 1 2; do
for c in C3 C1; do
BARGS="--config $c --size 512 512 512"
run "default     " X=1
run "order0      " BFD_RUN_ORDER=0
run "zrun32      " BFD_ZRUN=32
BARGS="--config $c --size 512 512 512 --variant 4"
run "variant4    " X=1
done; done

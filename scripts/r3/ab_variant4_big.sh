# fused fluid time step (variant 4, second copies of V / Szz / Rzz, not yet placed by region) against the default on the large volumes
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_v4; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -3
for c in C5 C4; do for v in 0 4; do
  timeout 900 python bench.py --config $c --scaling strong --no-cpu-baseline --no-kernel-pass --variant $v > $O/big_${c}_v$v.json 2> $O/big_${c}_v$v.err
  python - $O/big_${c}_v$v.json $c $v <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], 'variant', sys.argv[3], round(d['value']), round(d['ms_per_step'],3), d['config'].get('tiles_rank0'), d['config'].get('array_placement','')[:60])
except Exception as e: print(sys.argv[2], sys.argv[3], 'failed', e)
PY
done; done

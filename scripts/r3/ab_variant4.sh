# the fused fluid time step (variant 4) against the default on the round-3 binary: C1 (water) / C2 medium / C3 at 512^3, fused z-runs of 16 / 32 / 64 planes
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_v4; mkdir -p $O
run() { # name, env, args
  local n=$1; shift; local e=$1; shift
  env $e timeout 600 python bench.py --no-cpu-baseline --no-shear-workload --no-kernel-pass "$@" > $O/$n.json 2>$O/$n.err
  python - $O/$n.json "$n" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d['value']), round(d['ms_per_step'],4), d['config'].get('tiles_rank0'), d['config'].get('array_placement','')[:80])
except Exception as e: print(sys.argv[2], 'failed', e)
PY
}
for cfg in C1 C2 C3; do
  run ${cfg}_v0 "A=1" --config $cfg --size 512 512 512 --variant 0
  for z in 16 32 64; do run ${cfg}_v4_z$z "BFD_FUSED_ZRUN=$z" --config $cfg --size 512 512 512 --variant 4; done
done

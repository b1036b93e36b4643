# same box: default / fused fluid time step (variant 4) x arrays placed by region or not; C1 and C3 at 512^3, C5
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_v4p; mkdir -p $O
run() { local n=$1; shift; local pl=$1; shift
  BFD_PLACEMENT=$pl timeout 900 python bench.py --no-cpu-baseline --no-kernel-pass --no-shear-workload "$@" > $O/$n.json 2> $O/$n.err
  python - $O/$n.json $n <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d['value']), round(d['ms_per_step'],3), d['config'].get('tiles_rank0',{}).get('fused_fluid'), d['config'].get('array_placement','')[:330])
except Exception as e: print(sys.argv[2], 'failed', e)
PY
}
for i in 1 2; do for v in 0 4; do for pl in 1 0; do
  run C1_v${v}_pl${pl}_$i $pl --config C1 --size 512 512 512 --variant $v
  run C3_v${v}_pl${pl}_$i $pl --config C3 --variant $v
done; done; done
for v in 0 4; do for pl in 1 0; do run C5_v${v}_pl${pl} $pl --config C5 --scaling strong --variant $v; done; done

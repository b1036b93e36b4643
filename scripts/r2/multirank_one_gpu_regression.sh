cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_mr; mkdir -p $O
BFD_PLACEMENT_VERBOSE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29713 bench.py --gpus 2 --debug-gloo-shared-gpu --steps 20 --warmup 5 > $O/c3_weak_2.json 2> $O/c3_weak_2.err; grep "placement: kept" $O/c3_weak_2.err; tail -2 $O/c3_weak_2.err
python - <<'PY'
import json
for l in open('gpurun_out/r2_mr/c3_weak_2.json'):
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(round(d['value']), d['n_gpus'], d['scaling'], d['config']['halo_exchange'], d['config'].get('halo_exchange_check'), d['config']['workload'][:80])
PY

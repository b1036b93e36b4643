cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_run13; mkdir -p $O
summ() { python - "$1" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    print(sys.argv[1].split('/')[-1], round(d['value']), round(d['ms_per_step'],4), d['scaling'], d['n_gpus'], d['config']['halo_exchange'], d['config'].get('halo_exchange_check'), {k:round(v['avg_launch_ms'],3) for k,v in d.get('roofline_kernels',{}).items()})
except Exception as e: print(sys.argv[1], 'failed', e)
PY
}
# multi-rank code path on one GPU (gloo, host-staged halos): strong scaling of C4 and weak scaling of C3-sized slabs
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 2 --config C4 --scaling strong --debug-gloo-shared-gpu --steps 20 --warmup 5 --no-kernel-pass > $O/c4_strong_2.json 2> $O/c4_strong_2.err; summ $O/c4_strong_2.json; tail -3 $O/c4_strong_2.err
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29712 bench.py --gpus 4 --config C5 --scaling strong --debug-gloo-shared-gpu --steps 10 --warmup 3 --no-kernel-pass > $O/c5_strong_4.json 2> $O/c5_strong_4.err; summ $O/c5_strong_4.json; tail -3 $O/c5_strong_4.err
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29713 bench.py --gpus 2 --debug-gloo-shared-gpu --steps 20 --warmup 5 --size 512 512 256 --no-kernel-pass > $O/c3_weak_2.json 2> $O/c3_weak_2.err; summ $O/c3_weak_2.json; tail -3 $O/c3_weak_2.err
# fused fluid time step (variant 4) after the addressing change
for rep in 1 2; do
python bench.py --no-cpu-baseline --no-shear-workload --steps 200 --warmup 30 > $O/c3_v0_$rep.json 2>/dev/null; summ $O/c3_v0_$rep.json
python bench.py --no-cpu-baseline --no-shear-workload --steps 200 --warmup 30 --variant 4 > $O/c3_v4_$rep.json 2>/dev/null; summ $O/c3_v4_$rep.json
python bench.py --config C1 --size 512 512 512 --no-cpu-baseline --steps 200 --warmup 30 > $O/c1_v0_$rep.json 2>/dev/null; summ $O/c1_v0_$rep.json
python bench.py --config C1 --size 512 512 512 --no-cpu-baseline --steps 200 --warmup 30 --variant 4 > $O/c1_v4_$rep.json 2>/dev/null; summ $O/c1_v4_$rep.json
done

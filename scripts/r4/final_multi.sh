# round 4, final binary: the two multi-device invocations of the bench on the 1-GPU box
#  (a) the driver's launcher form with 2 ranks sharing the GPU (gloo, --debug-gloo-shared-gpu), driver's flags otherwise
#  (b) launcher-free --gpus 2 (bfd_group, ordinals repeated: "emulated")
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_multi; mkdir -p $O
S=$(date +%s)
timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29713 bench.py --gpus 2 --debug-gloo-shared-gpu --steps 20 --warmup 5 > $O/torchrun2.json 2> $O/torchrun2.err; echo "torchrun rc=$? $(( $(date +%s)-S )) s"
S=$(date +%s)
timeout 1500 python bench.py --gpus 2 --steps 20 --warmup 5 > $O/gpus2.json 2> $O/gpus2.err; echo "launcher-free rc=$? $(( $(date +%s)-S )) s"
python - <<'PY'
import json
for n in ('torchrun2', 'gpus2'):
    try:
        d = json.loads(open('gpurun_out/r4_multi/%s.json' % n).read().strip().split('\n')[-1])
        print(n, round(d['value']), d['scaling'], d['n_gpus'], d['config'].get('workload', '')[:80])
        for k in ('extra_strong_c5', 'group_strong_c3', 'secondary_weak_c3', 'group_check'):
            v = d.get(k)
            if isinstance(v, dict): print('   ', k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a in ('value', 'error', 'equals_single_domain', 'ms_per_step', 'emulated', 'host_issue_ms_per_step')})
            elif v is not None: print('   ', k, v)
        print('    halo checks:', d['config'].get('halo_exchange_check'), d['config'].get('halo_exchange_vs_single_domain'), 'emulated' , d.get('emulated'))
    except Exception as e:
        print(n, 'failed', e)
PY
tail -n 3 $O/torchrun2.err; tail -n 3 $O/gpus2.err

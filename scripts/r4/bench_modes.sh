# round 4: the bench in its three invocations on the 1-GPU box: default (N=1), launcher-free --gpus 2 (bfd_group, emulated), and the new C3 oracle test
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_bench; mkdir -p $O
timeout 900 python -m pytest tests/test_configs_gpu.py -x -q -k "c3_full_size" -s > $O/pytest_c3.txt 2>&1; tail -4 $O/pytest_c3.txt
timeout 900 python bench.py > $O/default.json 2> $O/default.err; echo "default rc=$?"; tail -c 600 $O/default.err
timeout 1500 python bench.py --gpus 2 --steps 20 --warmup 5 > $O/gpus2.json 2> $O/gpus2.err; echo "gpus2 rc=$?"; tail -c 600 $O/gpus2.err
python - <<'PY'
import json
for n in ('default','gpus2'):
    try:
        d=json.load(open('gpurun_out/r4_bench/%s.json'%n))
        print(n, round(d['value']), d['ms_per_step'], d.get('windows'), d.get('emulated'), {k:(v.get('value') if isinstance(v,dict) else None) for k,v in d.items() if isinstance(v,dict) and 'value' in v})
        print('  group_check', d.get('group_check'), 'cpu', d.get('cpu_baseline'))
    except Exception as e: print(n,'failed',e)
PY

# round 2, run 2: refocus + C5 tests, shear-kernel XCD order effect on the C2-medium 512^3 workload (bench + FETCH/WRITE passes)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_run2; mkdir -p $O
timeout 2400 python -m pytest tests/test_refocus_gpu.py tests/test_configs_gpu.py -m gpu -q -x -k "refocus or c5_1024" --durations=5 -o faulthandler_timeout=1500 > $O/tests.log 2>&1; tail -12 $O/tests.log
timeout 600 python bench.py --config C2 --size 512 512 512 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; tail -c 300 $O/bench_c2.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r2_run2/bench_c2.json'))
print('C2-512', round(d['value']), d['ms_per_step'])
for k,v in d['roofline_kernels'].items(): print(k, round(v['avg_launch_ms'],4), round(v['frac'],3))
PY
export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_$grp -- python3 bench.py --config C2 --size 512 512 512 --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-pass --no-steady-warmup > $O/pmc_$grp.log 2>&1
done
mkdir -p $O/pmc; mv $O/pmc_FETCH_SIZE $O/pmc/p1; mv $O/pmc_WRITE_SIZE $O/pmc/p2
python3 scripts/pmc_summary.py $O/pmc | grep -E "^==|HBM"

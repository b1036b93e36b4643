# on the GPU box: tests, default bench, rocprofv3 kernel stats and PMC traffic of the same command
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout 600 python -m pytest tests -m gpu -q -x -o faulthandler_timeout=240 > gpurun_out/final/tests.log 2>&1; tail -3 gpurun_out/final/tests.log
timeout 600 python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err; python -c "
import json; d=json.load(open('gpurun_out/final/bench.json')); print('bench', round(d['value']), d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d.get('dense_reference',{}).get('value'))"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/ktrace -o k -- python3 bench.py --no-cpu-baseline --no-dense-reference > gpurun_out/final/bench_prof.json 2>/dev/null
ls gpurun_out/final/ktrace/* | head
bash scripts/pmc_passes.sh final --no-dense-reference > gpurun_out/final/pmc.log 2>&1; tail -30 gpurun_out/final/pmc.log | grep -E "^==|HBM"

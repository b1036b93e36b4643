# round-3 evidence: default bench line, kernel-trace stats and PMC traffic of the two 512^3 workloads of the bench
# (C3 = CT-like all-fluid skull, the metric's config; C2 medium = water / cortical bone with shear / brain), next-row kernels
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_prof; mkdir -p $O
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace_c3 -o k -- python3 bench.py --no-cpu-baseline --no-shear-workload --no-next-rows > $O/bench_c3_under_rocprof.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace_c2 -o k -- python3 bench.py --config C2 --size 512 512 512 --no-cpu-baseline --no-next-rows > $O/bench_c2_under_rocprof.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace_next -o k -- python3 scripts/next_rows_bench.py > $O/next_rows_under_rocprof.txt 2>/dev/null
TRAFFIC_KEY=C3_512x512x512_variant0 bash scripts/pmc_passes.sh r3_c3 > $O/pmc_c3.log 2>&1
TRAFFIC_KEY=C2_512x512x512_variant0 bash scripts/pmc_passes.sh r3_c2 --config C2 --size 512 512 512 > $O/pmc_c2.log 2>&1
grep -E "^==|HBM" $O/pmc_c3.log $O/pmc_c2.log | head -40
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3_prof/bench_default.json'))
print(round(d['value']), d['ms_per_step'], d['roofline']['kernel'], round(d['roofline']['frac'],3))
print(d['config']['array_placement'])
s=d['shear_workload']; print(round(s['value']), s['ms_per_step'])
for k,v in s['roofline_kernels'].items(): print('  ',k, round(v['avg_launch_ms'],4), round(v['frac'],3))
print(d['cpu_baseline'])
PY

# round-4 evidence: default bench line, the launcher-free --gpus 2 line, kernel-trace stats and PMC traffic of the 512^3 workloads
# (C3 = CT-like all-fluid skull, the metric's config; C2 medium = water / cortical bone with shear / brain; C1 and C3 with the fused step)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_prof; mkdir -p $O
timeout 900 python bench.py > $O/bench_default_n1.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace_c3 -o k -- python3 bench.py --no-cpu-baseline --no-shear-workload --no-next-rows --no-group > $O/bench_c3_under_rocprof.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace_c2 -o k -- python3 bench.py --config C2 --size 512 512 512 --no-cpu-baseline --no-next-rows --no-group > $O/bench_c2medium_under_rocprof.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace_c3v4 -o k -- python3 bench.py --variant 4 --no-cpu-baseline --no-shear-workload --no-next-rows --no-group > $O/bench_c3_variant4_under_rocprof.json 2>/dev/null
TRAFFIC_KEY=C3_512x512x512_variant0 bash scripts/pmc_passes.sh r4_c3 --no-group > $O/pmc_c3.log 2>&1
TRAFFIC_KEY=C2_512x512x512_variant0 bash scripts/pmc_passes.sh r4_c2 --config C2 --size 512 512 512 --no-group > $O/pmc_c2.log 2>&1
TRAFFIC_KEY=C3_512x512x512_variant4 bash scripts/pmc_passes.sh r4_c3v4 --variant 4 --no-group > $O/pmc_c3v4.log 2>&1
timeout 1500 python bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_gpus2_launcher_free_emulated.json 2> $O/gpus2.err
grep -E "^==|HBM" $O/pmc_c3.log $O/pmc_c2.log $O/pmc_c3v4.log | head -60
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4_prof/bench_default_n1.json'))
print(round(d['value']), d['ms_per_step'], d['windows']['ms_per_step'], d['roofline']['kernel'], round(d['roofline']['frac'],3))
s=d['shear_workload']; print('shear', round(s['value']), s['ms_per_step'], round(s['roofline_step']['frac'],3))
for k,v in s['roofline_kernels'].items(): print('  ',k, round(v['avg_launch_ms'],4), round(v['frac'],3))
print(d['cpu_baseline']); print(d['group_one_slab']['value'], d['next_rows'])
g=json.load(open('gpurun_out/r4_prof/bench_gpus2_launcher_free_emulated.json')); print('gpus2', round(g['value']), g['emulated'], g['group_check'], g['secondary_weak_c3'].get('value'))
PY

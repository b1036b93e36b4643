# round-3 refresh for the two large configs, each whole on one GPU: kernel-trace stats and the HBM-byte counters per kernel class
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_prof45; rm -rf $O gpurun_out/pmc_r3_c4 gpurun_out/pmc_r3_c5; mkdir -p $O
for c in C4 C5; do
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace_$c -o k -- python3 bench.py --config $c --scaling strong --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-pass --no-next-rows > $O/bench_${c}_under_rocprof.json 2>/dev/null
done
PMC_TRAFFIC_ONLY=1 PMC_TIMEOUT=600 TRAFFIC_KEY=C4_512x512x1024_variant0 bash scripts/pmc_passes.sh r3_c4 --config C4 --scaling strong > $O/pmc_c4.log 2>&1
PMC_TRAFFIC_ONLY=1 PMC_TIMEOUT=900 TRAFFIC_KEY=C5_1024x1024x1024_variant0 bash scripts/pmc_passes.sh r3_c5 --config C5 --scaling strong > $O/pmc_c5.log 2>&1
grep -E "^==|HBM" $O/pmc_c4.log $O/pmc_c5.log | head -40
cat gpurun_out/pmc_r3_c4/traffic.json gpurun_out/pmc_r3_c5/traffic.json
find $O gpurun_out/pmc_r3_c4 gpurun_out/pmc_r3_c5 -name "*.csv" -size +3M -delete

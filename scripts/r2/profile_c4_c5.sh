# round-2 evidence for the two multi-GPU configs of the north star, each whole on one GPU: bench line, kernel-trace stats and
# the HBM-byte counters (FETCH_SIZE, WRITE_SIZE, TCC hit/miss) per kernel class.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_prof45; mkdir -p $O
for c in C4 C5; do
  timeout 900 python bench.py --config $c --scaling strong --steps 40 --warmup 10 --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; tail -c 200 $O/bench_$c.err
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace_$c -o k -- python3 bench.py --config $c --scaling strong --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-pass > $O/bench_${c}_under_rocprof.json 2>/dev/null
done
PMC_TRAFFIC_ONLY=1 PMC_TIMEOUT=600 TRAFFIC_KEY=C4_512x512x1024_variant0 bash scripts/pmc_passes.sh r2_c4 --config C4 --scaling strong > $O/pmc_c4.log 2>&1
PMC_TRAFFIC_ONLY=1 PMC_TIMEOUT=900 TRAFFIC_KEY=C5_1024x1024x1024_variant0 bash scripts/pmc_passes.sh r2_c5 --config C5 --scaling strong > $O/pmc_c5.log 2>&1
grep -E "^==|HBM" $O/pmc_c4.log $O/pmc_c5.log | head -40
find $O -name "*.csv" -size +3M -delete
ls $O/ktrace_C5 $O/ktrace_C4 2>/dev/null | head

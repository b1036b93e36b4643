# round 4, last binary: kernel-trace stats of the C3 and C2-medium 512^3 workloads again (the PMC passes of scripts/r4/profile.sh stay: the kernels' bodies did not change after them)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_prof_final; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace_c3 -o k -- python3 bench.py --no-cpu-baseline --no-shear-workload --no-next-rows --no-group --no-production-schedule > $O/bench_c3_under_rocprof.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace_c2 -o k -- python3 bench.py --config C2 --size 512 512 512 --no-cpu-baseline --no-next-rows --no-group --no-production-schedule > $O/bench_c2medium_under_rocprof.json 2>/dev/null
for c in c3 c2; do echo "== $c"; head -8 $O/ktrace_$c/k_kernel_stats.csv | cut -c1-200; done

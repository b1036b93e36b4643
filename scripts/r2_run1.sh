# round 2, first GPU call: box facts, GPU tests (without C5), default bench, kernel stats + PMC of the C2-medium 512^3 workload (baseline)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_run1; mkdir -p $O
(free -g; cat /sys/fs/cgroup/memory.max; cat /sys/fs/cgroup/cpu.max; nproc; df -h /tmp | tail -1) > $O/box.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -q -x -k "not c5_1024" --durations=15 -o faulthandler_timeout=600 > $O/tests.log 2>&1; tail -25 $O/tests.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace_c2 -o k -- python3 bench.py --config C2 --size 512 512 512 --steps 100 --warmup 20 --no-cpu-baseline --no-kernel-pass > $O/bench_c2_prof.json 2>$O/bench_c2_prof.err
TRAFFIC_KEY=C2_512x512x512_variant0 bash scripts/pmc_passes.sh r2_c2_before --config C2 --size 512 512 512 --no-kernel-pass --no-steady-warmup > $O/pmc.log 2>&1; grep -E "^==|HBM" $O/pmc.log | head -40

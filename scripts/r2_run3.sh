# round 2, run 3: parity of the class-predicated solid kernels, C2-medium / C3 / C5 timings, FETCH/WRITE of C2-medium
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_run3; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x -k "not c5_1024 and not c4_h317 and not c2_full" --durations=5 -o faulthandler_timeout=600 > $O/tests.log 2>&1; tail -8 $O/tests.log
timeout 600 python bench.py --config C2 --size 512 512 512 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; tail -c 300 $O/bench_c2.err
timeout 600 python bench.py --no-cpu-baseline --no-shear-workload > $O/bench_c3.json 2> $O/bench_c3.err; tail -c 300 $O/bench_c3.err
timeout 900 python bench.py --config C5 --scaling strong --steps 60 --warmup 10 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; tail -c 300 $O/bench_c5.err
python - <<'PY'
import json
for n in ('c2','c3','c5'):
    try:
        d=json.load(open('gpurun_out/r2_run3/bench_%s.json'%n))
        print(n, round(d['value']), round(d['ms_per_step'],4), d['config']['tiles_rank0'])
        for k,v in d['roofline_kernels'].items(): print('   ',k, round(v['avg_launch_ms'],4), round(v['frac'],3), round(v['algorithmic_bytes_per_launch']/1e9,3))
    except Exception as e: print(n, 'failed', e)
PY
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_$grp -- python3 bench.py --config C2 --size 512 512 512 --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-pass --no-steady-warmup > $O/pmc_$grp.log 2>&1
done
mkdir -p $O/pmc; mv $O/pmc_FETCH_SIZE $O/pmc/p1; mv $O/pmc_WRITE_SIZE $O/pmc/p2
python3 scripts/pmc_summary.py $O/pmc | grep -E "^==|HBM"

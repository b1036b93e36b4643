# round 5: what the waves of each kernel spend their cycles on (SQ counters, one pass; quad-cycle units, summed over the chip) -- C2's medium at 512^3
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_sq; mkdir -p $O
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $O/p1 -- python3 bench.py --steps 4 --warmup 2 --windows 1 --no-kernel-pass --no-steady-warmup --no-cpu-baseline --no-shear-workload --no-next-rows --no-group --no-production-schedule --no-wide-placement --no-strong-c5 --config C2 --size 512 512 512 > $O/p1.log 2>&1
python3 scripts/pmc_summary.py $O x > $O/summary.txt
grep -A14 "== .*\(stress_fluid\|velocity_fluid\|velocity_solid<true, false\|stress_shear\)" $O/summary.txt | grep "==\|SQ_\|dur_us"
find $O -name "*.csv" -size +300k -delete; find $O -name "*.db" -delete

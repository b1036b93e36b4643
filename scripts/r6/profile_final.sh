# round 6, final binary: kernel-trace stats + the three PMC passes (FETCH_SIZE / WRITE_SIZE / L2 hits, each its own run) of the four workloads the
# bench prices -- C3 512^3 (metric config), C2's medium at 512^3 (shear), C4 512x512x1024 and C5 1024^3 whole on one GPU.
# usage: bash scripts/r6/profile_final.sh [c3 c2 c4 c5]      then, here: python scripts/r6/make_traffic.py
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_prof; mkdir -p $O
COMMON="--no-cpu-baseline --no-shear-workload --no-next-rows --no-group --no-production-schedule --no-wide-placement --no-strong-c5"
for w in ${@:-c3 c2 c4 c5}; do
  case $w in
    c3) A="";;
    c2) A="--config C2 --size 512 512 512";;
    c4) A="--config C4 --scaling strong";;
    c5) A="--config C5 --scaling strong --steps 40 --warmup 10 --windows 1";;
  esac
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace_$w -o k -- python3 bench.py $COMMON $A > $O/bench_${w}_under_rocprof.json 2>$O/bench_${w}.err
  cp $O/ktrace_$w/k_kernel_stats.csv $O/kernel_stats_$w.csv 2>/dev/null
  mkdir -p $O/pmc_$w; i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout 600 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_$w/p$i -- python3 bench.py --steps 4 --warmup 2 --windows 1 --no-kernel-pass --no-steady-warmup $COMMON $A > $O/pmc_$w/p$i.log 2>&1
  done
  python3 scripts/pmc_summary.py $O/pmc_$w $w > $O/pmc_summary_$w.txt
  echo "== $w"; head -7 $O/kernel_stats_$w.csv | cut -c1-160; grep "==\|HBM" $O/pmc_summary_$w.txt | head -20
  find $O/ktrace_$w $O/pmc_$w -name "*.csv" -size +300k -delete; find $O -name "*.db" -delete
done

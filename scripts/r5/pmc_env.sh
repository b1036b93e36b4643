# round 5: HBM-side traffic (FETCH_SIZE / WRITE_SIZE / L2 hits) of a bench run under environment settings
# usage: ENVS="A=1 B=2" bash scripts/r5/pmc_env.sh <outtag> [bench args]     (the variables must be exported by the caller: rocprofv3 wants the program right after --)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=$1; shift
mkdir -p gpurun_out/pmc_$tag
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" ${PMC_EXTRA}; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmc_$tag/p$i -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-pass --no-steady-warmup --no-shear-workload --no-next-rows --no-group --no-production-schedule "$@" > gpurun_out/pmc_$tag/p$i.log 2>&1
done
python3 scripts/pmc_summary.py gpurun_out/pmc_$tag x > gpurun_out/pmc_$tag/summary.txt
grep -B1 -A14 "== .*\(stress\|velocity\)" gpurun_out/pmc_$tag/summary.txt | grep "==\|dur_us\|HBM"
find gpurun_out/pmc_$tag -name "*.csv" -size +200k -delete; find gpurun_out/pmc_$tag -name "*.db" -delete

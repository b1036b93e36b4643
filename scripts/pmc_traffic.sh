# HBM-side traffic only (3 passes): bash scripts/pmc_traffic.sh <tag> [bench args]
export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out/pmc_$tag
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmc_$tag/p$i -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-dense-reference "$@" > gpurun_out/pmc_$tag/p$i.log 2>&1
done
python3 scripts/pmc_summary.py gpurun_out/pmc_$tag > gpurun_out/pmc_$tag/summary.txt
grep -E "^==|HBM bytes|TCC_|dur_us" gpurun_out/pmc_$tag/summary.txt

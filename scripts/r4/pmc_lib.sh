# round 4: HBM-side traffic (FETCH_SIZE / WRITE_SIZE / L2 hits) of a bench run with an experiment build of the library
# usage: LIBTAG=<tag|base> bash scripts/r4/pmc_lib.sh <outtag> [bench args]
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
tag=$1; shift
lib=babelbrain_amd/libbabelfdtd_hip.so; [ "${LIBTAG:-base}" != base ] && lib=babelbrain_amd/libbabelfdtd_hip_$LIBTAG.so
export BABELFDTD_HIP_LIB=$PWD/$lib
mkdir -p gpurun_out/pmc_$tag
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" ${PMC_EXTRA}; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmc_$tag/p$i -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-pass --no-steady-warmup --no-shear-workload --no-next-rows "$@" > gpurun_out/pmc_$tag/p$i.log 2>&1
done
python3 scripts/pmc_summary.py gpurun_out/pmc_$tag x > gpurun_out/pmc_$tag/summary.txt
grep -A12 "== .*${KERNEL:-fused}" gpurun_out/pmc_$tag/summary.txt | grep "==\|FETCH\|WRITE\|TCC\|dur_us\|HBM"

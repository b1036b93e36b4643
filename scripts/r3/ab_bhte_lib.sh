# BHTE kernels of the product library against a variant build (make TAG=<name>), alternating, 384^3; then the BHTE tests on the product library
cd $GRAFT_REPO_ROOT; T=$1
for i in 1 2 3; do for t in "" $T; do
  L=""; [ -n "$t" ] && L=$GRAFT_REPO_ROOT/babelbrain_amd/libbabelfdtd_hip_$t.so
  BABELFDTD_HIP_LIB=$L timeout 300 python scripts/next_rows_bench.py 2>&1 | grep 'two steps' | cut -c1-170 | sed "s/^/${t:-product}: /"
done; done
timeout 600 python -m pytest tests/test_bhte_gpu.py -m gpu -q 2>&1 | grep -E "passed|failed" | tail -2

# Rayleigh kernel with 1 / 2 / 4 field points per lane (BFD_RAYLEIGH_PPL; default: by the number of points): source plane and volume rates, then the tests
cd $GRAFT_REPO_ROOT
for i in 1 2; do for v in default 1 2 4; do
  E=""; [ $v != default ] && E="BFD_RAYLEIGH_PPL=$v"
  env $E timeout 300 python scripts/next_rows_bench.py 2>&1 | grep 'Rayleigh' | cut -c1-150 | sed "s/^/ppl $v: /"
done; done
timeout 900 python -m pytest tests/test_rayleigh_gpu.py tests/test_rayleigh_study_gpu.py -m gpu -q 2>&1 | grep -E "passed|failed" | tail -3

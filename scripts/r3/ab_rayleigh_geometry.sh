# Rayleigh kernel: Newton step on the square root (make TAG=g1 EXTRA=-DBFD_RAYLEIGH_GEOMETRY=1) against refining the reciprocal square root: rate, error, study rows
cd $GRAFT_REPO_ROOT
for i in 1 2; do for t in "" g2 g1; do
  L=""; [ -n "$t" ] && L=$GRAFT_REPO_ROOT/babelbrain_amd/libbabelfdtd_hip_$t.so
  BABELFDTD_HIP_LIB=$L timeout 300 python scripts/next_rows_bench.py 2>&1 | grep 'Rayleigh' | cut -c1-150 | sed "s/^/${t:-g0}: /"
done; done
BABELFDTD_HIP_LIB=$GRAFT_REPO_ROOT/babelbrain_amd/libbabelfdtd_hip_g2.so timeout 900 python -m pytest tests/test_rayleigh_gpu.py tests/test_rayleigh_study_gpu.py tests/test_refocus_gpu.py -m gpu -q -s 2>&1 | grep -E "rel L2|passed|failed" | tail -8

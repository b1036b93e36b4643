# Rayleigh kernel, source loop unrolled by 1 / 2 (shipped) / 4 (make TAG=u1 EXTRA=-DBFD_RAYLEIGH_UNROLL=1, ...)
cd $GRAFT_REPO_ROOT
for i in 1 2; do for t in "" u1 u4; do
  L=""; [ -n "$t" ] && L=$GRAFT_REPO_ROOT/babelbrain_amd/libbabelfdtd_hip_$t.so
  BABELFDTD_HIP_LIB=$L timeout 300 python scripts/next_rows_bench.py 2>&1 | grep 'Rayleigh' | cut -c1-150 | sed "s/^/${t:-u2}: /"
done; done

# round 6 experiment: bhte_stepNg with two cells per thread (1024 threads per workgroup, -DGN_NC=2; LIBS names the builds) against four. usage: LIBS="_nc2w4" bash scripts/r6/bhte_nc_ab.sh
cd $GRAFT_REPO_ROOT
for n in 320 512; do for on in 0 200; do for st in 3 4; do
  for lib in "" ${LIBS:-_nc2}; do
    BABELFDTD_HIP_LIB=$PWD/babelbrain_amd/libbabelfdtd_hip$lib.so BFD_BHTE_STEPS=$st python scripts/r4/bhte_bench.py $n 200 $on 2>&1 | grep "^BHTE" | sed "s/^BHTE/lib=${lib:-base} steps=$st on=$on BHTE/" | cut -c1-120
  done
done; done; done

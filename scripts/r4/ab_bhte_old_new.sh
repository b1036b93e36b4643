# same-box A/B: bhte_step2 (round 3, BFD_BHTE_KERNEL=1) against bhte_step2g (default), alternating
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_bhte; mkdir -p $O
for n in ${SIZES:-384 512 256}; do for z in ${ZRUNS:-0 24}; do for k in 1 0 1 0; do
  BFD_BHTE_ZRUN=$z BFD_BHTE_KERNEL=$k timeout 300 python scripts/r4/bhte_bench.py $n ${STEPS:-200 100} 2>&1 | tail -1
done; done; done | tee $O/old_new_${OUT:-run}.txt

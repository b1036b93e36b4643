# round 4: bhte_step2g (GLOBAL loads that stay in flight, 64 x 24 tiles, column layout) against bhte_step2 (round 3): parity tests in both modes, then rates
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_bhte; mkdir -p $O
timeout 600 python -m pytest tests/test_bhte_gpu.py -x -q > $O/pytest_g.txt 2>&1; echo "new kernel: $(tail -1 $O/pytest_g.txt)"
BFD_BHTE_KERNEL=1 timeout 600 python -m pytest tests/test_bhte_gpu.py -x -q > $O/pytest_old.txt 2>&1; echo "old kernel: $(tail -1 $O/pytest_old.txt)"
for n in ${SIZES:-384 256 512}; do
  for k in 1 0 1 0; do BFD_BHTE_KERNEL=$k timeout 300 python scripts/r4/bhte_bench.py $n 200 100 2>&1 | tail -1; done
done | tee $O/rates.txt
for z in ${ZRUNS:-8 12 16 24 32}; do BFD_BHTE_ZRUN=$z timeout 300 python scripts/r4/bhte_bench.py 384 200 100 2>&1 | tail -1; done | tee $O/zruns.txt

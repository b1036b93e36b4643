# two steps per launch (bhte_step2): parity tests, then throughput at 384^3 and 256x200x300: items handed out at run time or one workgroup per item, z-run lengths
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_bhte; mkdir -p $O
for w in 1 0; do BFD_BHTE_DYNAMIC=$w timeout 600 python -m pytest tests/test_bhte_gpu.py -m gpu -x -q 2>&1 | tail -1; done
for i in 1 2; do for w in 1 0; do for z in 8 12 16 24; do
  echo "dynamic $w zrun $z: $(BFD_BHTE_DYNAMIC=$w BFD_BHTE_ZRUN=$z timeout 300 python scripts/next_rows_bench.py 2>&1 | grep 'two steps')"
done; done; done | tee $O/sweep5.txt
for g in 512 768 1024 1536; do echo "dynamic grid $g zrun 12: $(BFD_BHTE_DYN_GRID=$g BFD_BHTE_ZRUN=12 timeout 300 python scripts/next_rows_bench.py 2>&1 | grep 'two steps')"; done | tee -a $O/sweep5.txt
timeout 300 python scripts/next_rows_bench.py 2>&1 | grep BHTE | tee -a $O/sweep5.txt

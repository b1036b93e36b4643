# quick A/B of BHTE builds: LIBS="tag1 tag2" (base = the product library), 384^3 and 512^3
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_bhte; mkdir -p $O
for tag in ${LIBS:-base}; do
  lib=babelbrain_amd/libbabelfdtd_hip.so; [ $tag != base ] && lib=babelbrain_amd/libbabelfdtd_hip_$tag.so
  export BABELFDTD_HIP_LIB=$PWD/$lib
  echo "== $tag: $(timeout 600 python -m pytest tests/test_bhte_gpu.py -x -q 2>&1 | tail -1)"
  for n in ${SIZES:-384 512}; do for z in ${ZRUNS:-0}; do
    BFD_BHTE_ZRUN=$z timeout 300 python scripts/r4/bhte_bench.py $n 200 100 2>&1 | tail -1
  done; done
done | tee $O/quick_${OUT:-run}.txt

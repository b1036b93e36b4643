# HBM bytes of stress_shear_sparse per launch for the list orders (BFD_SHEAR_ORDER 0 / 1 / 2), C5 and the shear medium at 512^3
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for cfg in "C5 --config C5 --scaling strong" "C2 --config C2 --size 512 512 512"; do
  set -- $cfg; name=$1; shift
  for so in 0 1 2; do
    rm -rf gpurun_out/pmc_so${so}_$name
    BFD_SHEAR_ORDER=$so PMC_TRAFFIC_ONLY=1 TRAFFIC_KEY=x bash scripts/pmc_passes.sh so${so}_$name "$@" > gpurun_out/pmc_so${so}_$name.log 2>&1
    echo "$name order $so: $(grep -A12 '== stress_shear_sparse' gpurun_out/pmc_so${so}_$name.log | grep -E 'HBM' | tr '\n' ' ')"
  done
done

# PMC passes for a bench workload; one counter group per pass (gfx950 slot limits), no trace domains beside them.
# usage: TRAFFIC_KEY=<config>_<N1>x<N2>x<nk>_variant<v> [PMC_TRAFFIC_ONLY=1] [PMC_TIMEOUT=seconds] bash scripts/pmc_passes.sh <tag> [bench args]
export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out/pmc_$tag
groups=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"
        "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"
        "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM"
        "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
        "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum")
[ -n "$PMC_TRAFFIC_ONLY" ] && groups=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum")    # the HBM bytes only (big grids)
i=0
for grp in "${groups[@]}"; do
  i=$((i+1))
  timeout ${PMC_TIMEOUT:-300} rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmc_$tag/p$i -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-pass --no-steady-warmup --no-shear-workload --no-next-rows "$@" > gpurun_out/pmc_$tag/p$i.log 2>&1
done
python3 scripts/pmc_summary.py gpurun_out/pmc_$tag ${TRAFFIC_KEY:-C3_512x512x512_variant0} > gpurun_out/pmc_$tag/summary.txt
cat gpurun_out/pmc_$tag/summary.txt

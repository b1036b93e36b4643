# Builds: ab/flat = `git archive <commit before the variant> babelbrain_amd/csrc include | tar -x -C ab/flat` + make there (ab/ is
# git-ignored and travels with the snapshot); variant libraries: apply scripts/r2/patches/*.patch, `make TAG=pin2 EXTRA=-DBFD_PIN_MODE=2` etc.
# Same box: FLAT accesses (previous commit, ab/flat) against global-address-space accesses with the 32-bit offset pinned by a
# volatile asm (product), a plain asm (pin2) or not at all (pin0: two thirds of the accesses keep 64-bit VGPR addresses).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_as; mkdir -p $O
summ() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], round(d['value']), round(d['ms_per_step'],4), {k:round(v['avg_launch_ms'],3) for k,v in d.get('roofline_kernels',{}).items()})
PY
}
L=$GRAFT_REPO_ROOT/babelbrain_amd
for rep in 1 2; do
  for v in product pin2 pin0 flat; do
    case $v in product) lib=$L/libbabelfdtd_hip.so;; flat) lib=$GRAFT_REPO_ROOT/ab/flat/babelbrain_amd/libbabelfdtd_hip.so;; *) lib=$L/libbabelfdtd_hip_$v.so;; esac
    BABELFDTD_HIP_LIB=$lib python bench.py --config C2 --size 512 512 512 --no-cpu-baseline --steps 200 --warmup 30 > $O/c2_${v}_$rep.json 2>/dev/null; summ $O/c2_${v}_$rep.json
    BABELFDTD_HIP_LIB=$lib python bench.py --no-cpu-baseline --steps 300 --warmup 30 > $O/c3_${v}_$rep.json 2>/dev/null; summ $O/c3_${v}_$rep.json
  done
done

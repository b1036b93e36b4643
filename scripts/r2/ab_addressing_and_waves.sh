cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_run10; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x -k "not c5_1024 and not c4_h317 and not c2_full and not rayleigh_study" -o faulthandler_timeout=600 > $O/tests.log 2>&1; tail -4 $O/tests.log
summ() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], round(d['value']), round(d['ms_per_step'],4), {k:round(v['avg_launch_ms'],3) for k,v in d.get('roofline_kernels',{}).items()})
PY
}
for rep in 1 2; do
for lib in default occ; do
  if [ $lib = occ ]; then export BABELFDTD_HIP_LIB=$PWD/babelbrain_amd/libbabelfdtd_hip_occ.so; else unset BABELFDTD_HIP_LIB; fi
  python bench.py --config C2 --size 512 512 512 --no-cpu-baseline --steps 200 --warmup 30 > $O/c2_${lib}_$rep.json 2>/dev/null; summ $O/c2_${lib}_$rep.json
  python bench.py --no-cpu-baseline --no-shear-workload --steps 200 --warmup 30 > $O/c3_${lib}_$rep.json 2>/dev/null; summ $O/c3_${lib}_$rep.json
done
done
unset BABELFDTD_HIP_LIB
(cd ab/r1 && timeout 600 python bench.py --config C2 --size 512 512 512 --steps 200 --warmup 150 --no-cpu-baseline --no-dense-reference) > $O/r1_c2.json 2>/dev/null; summ $O/r1_c2.json
(cd ab/r1 && timeout 600 python bench.py --steps 200 --warmup 150 --no-cpu-baseline --no-dense-reference) > $O/r1_c3.json 2>/dev/null; summ $O/r1_c3.json

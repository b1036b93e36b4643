cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_run12; mkdir -p $O
summ() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], round(d['value']), round(d['ms_per_step'],4), {k:round(v['avg_launch_ms'],3) for k,v in d.get('roofline_kernels',{}).items()})
PY
}
for rep in 1 2 3; do
for lib in default vs6; do
  if [ $lib = vs6 ]; then export BABELFDTD_HIP_LIB=$PWD/babelbrain_amd/libbabelfdtd_hip_vs6.so; else unset BABELFDTD_HIP_LIB; fi
  python bench.py --config C2 --size 512 512 512 --no-cpu-baseline --steps 200 --warmup 30 > $O/c2_${lib}_$rep.json 2>/dev/null; summ $O/c2_${lib}_$rep.json
done
done

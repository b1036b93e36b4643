cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_run6; mkdir -p $O
B="python bench.py --config C2 --size 512 512 512 --no-cpu-baseline --steps 200 --warmup 30"
summ() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], round(d['value']), round(d['ms_per_step'],4), {k:round(v['avg_launch_ms'],3) for k,v in d['roofline_kernels'].items()})
PY
}
$B > $O/base.json 2>/dev/null; summ $O/base.json
BFD_CONCURRENT=1 $B > $O/conc1.json 2>/dev/null; summ $O/conc1.json
BFD_CONCURRENT=2 $B > $O/conc2.json 2>/dev/null; summ $O/conc2.json
BFD_CONCURRENT=3 $B > $O/conc3.json 2>/dev/null; summ $O/conc3.json
export BABELFDTD_HIP_LIB=$PWD/babelbrain_amd/libbabelfdtd_hip_early.so
$B > $O/early.json 2>/dev/null; summ $O/early.json
BFD_CONCURRENT=1 $B > $O/early_conc1.json 2>/dev/null; summ $O/early_conc1.json
unset BABELFDTD_HIP_LIB
$B > $O/base2.json 2>/dev/null; summ $O/base2.json
for v in "--cases 9 36 63 --depth-mm 50" "--cases 9 36 63 --depth-mm 80" "--cases 9 36 63 --gap-vox 0.5" "--cases 9 36 63 --gap-vox 2" ; do echo "== $v"; timeout 600 python scripts/rayleigh_study_sweep.py $v 2>/dev/null | cut -c1-150; done
timeout 1500 python scripts/rayleigh_study_sweep.py --zadj 0 -10 --out $O/study_all.json > $O/study_all.log 2>&1; tail -3 $O/study_all.log

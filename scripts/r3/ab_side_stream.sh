# same box A/B: the absorbing-layer flavour of velocity_solid on a side stream (default) vs every launch on the one stream (BFD_SIDE_STREAM=0)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r3_side; mkdir -p $O
for i in 1 2 3; do
  for v in 1 0; do
    BFD_SIDE_STREAM=$v timeout 600 python bench.py --config C2 --size 512 512 512 --no-cpu-baseline > $O/c2_side${v}_$i.json 2>/dev/null
    python - $O/c2_side${v}_$i.json $v <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print('side',sys.argv[2], round(d['value']), d['ms_per_step'], {k:round(v['avg_launch_ms'],4) for k,v in d.get('roofline_kernels',{}).items()})
PY
  done
done
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_random_media_gpu.py -m gpu -x -q 2>&1 | tail -3

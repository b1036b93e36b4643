# Same box: the velocity half-step walks the run lists backwards (BFD_REVERSE=1; =2 also launches its fluid runs first), so
# that each kernel starts on the data the previous one touched last (L2 4 MB per XCD, memory-side cache 256 MB).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_rev; mkdir -p $O
BFD_REVERSE=1 timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_slab_gpu.py -m gpu -q -x > $O/tests_rev1.log 2>&1; grep -E "passed|failed" $O/tests_rev1.log
summ() { python - "$1" <<'PY'
import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1].split('/')[-1], round(d['value']), round(d['ms_per_step'],4), {k:round(v['avg_launch_ms'],3) for k,v in d.get('roofline_kernels',{}).items()})
PY
}
for rep in 1 2 3; do
  for r in 0 1 2; do
    BFD_REVERSE=$r python bench.py --no-cpu-baseline --steps 300 --warmup 30 > $O/c3_rev${r}_$rep.json 2>/dev/null; summ $O/c3_rev${r}_$rep.json
  done
done
for rep in 1 2; do
  for r in 0 1 2; do
    BFD_REVERSE=$r python bench.py --config C2 --size 512 512 512 --no-cpu-baseline --steps 200 --warmup 30 > $O/c2_rev${r}_$rep.json 2>/dev/null; summ $O/c2_rev${r}_$rep.json
  done
done

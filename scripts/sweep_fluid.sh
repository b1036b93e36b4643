mkdir -p gpurun_out
for w in 8 6 5 4; do
  rm -f babelbrain_amd/csrc/bfd_kernels_v2.o
  make -C babelbrain_amd/csrc -s EXTRA="-DFLUID_WAVES_PER_SIMD=$w" > /dev/null 2>&1
  timeout 300 python bench.py --steps 30 --warmup 4 --no-cpu-baseline > gpurun_out/sweepf_w$w.json 2>/dev/null
  python - <<PY
import json
d=json.load(open('gpurun_out/sweepf_w$w.json'))
print('fluid w=$w value %.0f stress %.3f ms vel %.3f ms step %.3f ms' % (d['value'], d['roofline']['avg_launch_ms'], d['roofline_velocity']['avg_launch_ms'], d['roofline_step']['device_ms_per_step']))
PY
done

mkdir -p gpurun_out
for rep in 1 2; do
for z in 32 16 24; do
  rm -f babelbrain_amd/csrc/bfd_kernels_v2.o
  make -C babelbrain_amd/csrc -s EXTRA="-DBFD_ZCHUNK=$z" > /dev/null 2>&1
  timeout 300 python bench.py --steps 60 --warmup 6 --no-cpu-baseline > gpurun_out/s.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/s.json')); print('ZCHUNK=$z C3 value %.0f step %.3f' % (d['value'], d['roofline_step']['device_ms_per_step']), d['roofline']['avg_launch_ms'], d['roofline_other']['avg_launch_ms'])"
done
done

mkdir -p gpurun_out
for z in 32 16 8; do
  rm -f babelbrain_amd/csrc/bfd_kernels_v2.o
  make -C babelbrain_amd/csrc -s EXTRA="-DBFD_SUBZ=$z" > /dev/null 2>&1
  for rep in 1 2; do
  timeout 300 python bench.py --steps 60 --warmup 6 --no-cpu-baseline > gpurun_out/s.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/s.json')); print('SUBZ=$z C3 value %.0f step %.3f' % (d['value'], d['roofline_step']['device_ms_per_step']))"
  done
  timeout 300 python bench.py --config C2 --size 512 512 512 --steps 40 --warmup 4 --no-cpu-baseline > gpurun_out/s.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/s.json')); print('SUBZ=$z C2@512 value %.0f step %.3f' % (d['value'], d['roofline_step']['device_ms_per_step']), d['config']['tiles_rank0'])"
done

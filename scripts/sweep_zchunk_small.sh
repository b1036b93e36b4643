mkdir -p gpurun_out
for z in 32 16 8; do
  rm -f babelbrain_amd/csrc/bfd_kernels_v2.o
  make -C babelbrain_amd/csrc -s EXTRA="-DBFD_ZCHUNK=$z" > /dev/null 2>&1
  for cfg in "C2" "C1" "C3 --size 256 256 256" "C3 --size 320 320 384"; do
  timeout 300 python bench.py --config $cfg --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/s.json 2>/dev/null
  python -c "
import json; d=json.load(open('gpurun_out/s.json')); print('ZCHUNK=$z $cfg value %.0f step %.4f' % (d['value'], d['roofline_step']['device_ms_per_step']))"
  done
done

mkdir -p gpurun_out
for f in "" "-DABL_NOHALO" "-DABL_NOTABLE" "-DABL_NOBARRIER" "-DABL_NOMAT -DABL_NOTABLE" "-DABL_NOPML" "-DABL_NOHALO -DABL_NOTABLE -DABL_NOBARRIER -DABL_NOMAT -DABL_NOPML"; do
  rm -f babelbrain_amd/csrc/bfd_kernels_v2.o
  make -C babelbrain_amd/csrc -s EXTRA="$f" > /dev/null 2>&1
  timeout 300 python bench.py --steps 30 --warmup 4 --no-cpu-baseline > gpurun_out/abl.json 2>/dev/null
  python - <<PY
import json
d=json.load(open('gpurun_out/abl.json'))
print('%-70s stress %.3f ms vel %.3f ms step %.3f ms' % ("$f", d['roofline']['avg_launch_ms'], d['roofline_velocity']['avg_launch_ms'], d['roofline_step']['device_ms_per_step']))
PY
done

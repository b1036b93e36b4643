# HBM bytes of the Rayleigh / BHTE kernels (scripts/next_rows_bench.py), one counter group per pass
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_next; mkdir -p $O
i=0
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace -o k -- python3 scripts/next_rows_bench.py > $O/next_rows_under_rocprof.txt 2>/dev/null
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $O/p$i -- python3 scripts/next_rows_bench.py > $O/p$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/r3_next/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'bhte_step' in n or 'rayleigh_forward' in n:
            k = 'bhte_step2' if 'bhte_step2' in n else 'bhte_step' if 'bhte' in n else 'rayleigh_forward'
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print('==', k)
    for c, v in sorted(d.items()): print('   %-22s n=%4d avg=%.6g' % (c, len(v), sum(v) / len(v)))
    if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
        f = sum(d['FETCH_SIZE']) / len(d['FETCH_SIZE']); w = sum(d['WRITE_SIZE']) / len(d['WRITE_SIZE'])
        print('   HBM bytes/launch (gfx950: FETCH_SIZE x2 KB + WRITE_SIZE KB): read %.3f GB write %.3f GB total %.3f GB' % (2 * f * 1024 / 1e9, w * 1024 / 1e9, (2 * f + w) * 1024 / 1e9))
PY
cat $O/next_rows_under_rocprof.txt; grep -E 'bhte|rayleigh' $O/ktrace/*kernel_stats.csv $O/ktrace/*/*kernel_stats.csv 2>/dev/null | head

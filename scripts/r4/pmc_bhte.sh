# round 4: HBM-side bytes and SQ counters of the BHTE two-step kernels (bhte_step2g default, bhte_step2 with BFD_BHTE_KERNEL=1), 384^3, one group per pass
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_bhte_pmc; mkdir -p $O
export BFD_BHTE_ZRUN=${ZRUN:-0}
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM" ${PMC_EXTRA}; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $O/p$i -- python3 scripts/r4/bhte_bench.py 384 40 20 > $O/p$i.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace -o k -- python3 scripts/r4/bhte_bench.py 384 200 100 > $O/under_rocprof.txt 2>/dev/null
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('gpurun_out/r4_bhte_pmc/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'bhte_step2' in n and int(r['Grid_Size']) > 100000:
            k = n.replace('(anonymous namespace)::', '').split('(')[0]
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
            agg[k]['dur_us'].append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3)
for k, d in sorted(agg.items()):
    print('==', k)
    for c, v in sorted(d.items()): print('   %-22s n=%4d avg=%.6g' % (c, len(v), sum(v) / len(v)))
    if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
        f = sum(d['FETCH_SIZE']) / len(d['FETCH_SIZE']); w = sum(d['WRITE_SIZE']) / len(d['WRITE_SIZE'])
        print('   HBM bytes/launch (gfx950: FETCH_SIZE x2 KB + WRITE_SIZE KB): read %.3f GB write %.3f GB total %.3f GB' % (2 * f * 1024 / 1e9, w * 1024 / 1e9, (2 * f + w) * 1024 / 1e9))
PY
cat $O/under_rocprof.txt; grep -h bhte_step2 $O/ktrace/*kernel_stats.csv $O/ktrace/*/*kernel_stats.csv 2>/dev/null | head

# Counters of the fluid kernels on fast and slow placements of the same data (scripts/placement_probe.py builds 8 engines per
# process): one counter group per pass; per engine the mean duration and counter value of velocity_fluid / stress_fluid.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export BFD_PLACEMENT_TRIALS=0      # the raw placements, not the chosen ones
O=gpurun_out/placement_pmc; rm -rf $O; mkdir -p $O
i=0
for grp in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $O/p$i -- python3 scripts/placement_probe.py > $O/p$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/placement_pmc/p*/')):
    files = glob.glob(d + '**/*counter_collection.csv', recursive=True)
    if not files: print(d, 'no data'); continue
    rows = list(csv.DictReader(open(files[0])))
    for kern in ('velocity_fluid', 'stress_fluid'):
        per = collections.OrderedDict()
        for r in rows:
            if kern in r['Kernel_Name']:
                per.setdefault(r['Dispatch_Id'], {'dur': (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3})[r['Counter_Name']] = float(r['Counter_Value'])
        disp = list(per.values())
        n = len(disp) // 8
        if n == 0: continue
        names = [k for k in disp[0] if k != 'dur']
        print(d.split('/')[-2], kern, 'launches per engine', n)
        for e in range(8):
            blk = disp[e * n + 2:(e + 1) * n]
            print('   engine %d: %.1f us  ' % (e, sum(x['dur'] for x in blk) / len(blk)) + '  '.join('%s %.4g' % (c, sum(x.get(c, 0) for x in blk) / len(blk)) for c in names))
PY

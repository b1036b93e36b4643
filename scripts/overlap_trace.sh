cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for m in streams; do
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ovl_$m -o t -- python3 scripts/nccl_overlap_check.py $m > gpurun_out/ovl_$m.log 2>&1
python3 - $m <<'PY'
import csv, glob, sys
m = sys.argv[1]
f = glob.glob('gpurun_out/ovl_%s/**/*kernel_trace.csv' % m, recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:38], r.get('Queue_Id', ''), r.get('Stream_Id', '')) for r in csv.DictReader(open(f))]
rows.sort()
# last 40 kernels = ~2 steps
t0 = rows[-24][0]
print('==', m)
for a, b, n, q, st in rows[-24:]:
    print('%9.1f %9.1f %7.1f us  q%s s%s  %s' % ((a - t0) / 1e3, (b - t0) / 1e3, (b - a) / 1e3, q, st, n))
PY
done

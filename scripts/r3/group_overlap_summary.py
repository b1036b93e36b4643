"""Reads a rocprofv3 kernel trace + memory copy trace (csv) and reports, for the device-to-device copies (the halo planes of
the one-process split), how much of their time is covered by kernels running at the same time."""
import csv, glob, sys
root = sys.argv[1]
kern, cop = [], []
for f in glob.glob(root + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        kern.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
for f in glob.glob(root + '/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        d = r.get('Direction', '')
        cop.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), d))
d2d = [c for c in cop if 'DEVICE_TO_DEVICE' in c[2].upper() or 'DTOD' in c[2].upper()]
if not d2d:      # two slabs on one device: the runtime copies with a kernel of its own, which the kernel trace lists
    d2d = [(a, b, 'copyBuffer kernel') for a, b, n in kern if 'copyBuffer' in n]
    kern = [k for k in kern if 'copyBuffer' not in k[2]]
kern = [k for k in kern if 'stress' in k[2] or 'velocity' in k[2]]
kern.sort()
print('kernels %d, copies %d (device-to-device %d; directions: %s)' % (len(kern), len(cop), len(d2d), sorted(set(c[2] for c in cop))))
if not d2d:
    sys.exit(0)
# steady state: the last 60 % of the copies
d2d.sort()
d2d = d2d[int(0.4 * len(d2d)):]
tot = cov = 0
import bisect
starts = [k[0] for k in kern]
for s, e, _ in d2d:
    tot += e - s
    # union of kernel intervals intersected with [s, e]
    iv = []
    i = bisect.bisect_left(starts, s) - 64
    for k in kern[max(i, 0):]:
        if k[0] >= e:
            break
        if k[1] > s:
            iv.append((max(k[0], s), min(k[1], e)))
    iv.sort(); cur = None
    for a, b in iv:
        if cur is None or a > cur[1]:
            if cur: cov += cur[1] - cur[0]
            cur = [a, b]
        else:
            cur[1] = max(cur[1], b)
    if cur: cov += cur[1] - cur[0]
span = d2d[-1][1] - d2d[0][0]
print('steady state: %d copies, %.3f ms of copy time in a window of %.3f ms; %.1f %% of the copy time has a kernel running beside it'
      % (len(d2d), tot / 1e6, span / 1e6, 100.0 * cov / max(tot, 1)))
print('average copy %.1f us' % (tot / len(d2d) / 1e3))

"""Collates rocprofv3 --pmc CSVs (one directory per pass) into per-kernel averages."""
import collections
import csv
import glob
import sys

root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name']
        if 'stress' in name or 'velocity' in name or 'fused' in name or 'accumulate' in name or 'record_sensors' in name:
            short = name.split('(')[1].split(')')[-1] if False else name.replace('(anonymous namespace)::', '').split('(')[0]
            agg[short][r['Counter_Name']].append(float(r['Counter_Value']))
            agg[short]['VGPR'].append(float(r['VGPR_Count'])); agg[short]['SGPR'].append(float(r['SGPR_Count']))
            agg[short]['LDS'].append(float(r['LDS_Block_Size']))
            agg[short]['dur_us'].append((float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3)
import json
# bench.py's kernel classes (babelbrain_amd/_engine.py KERNEL_CLASSES) by kernel name
CLASS_OF = {'stress_fluid': 'stress_fluid', 'stress_normal_solid': 'stress_normal_solid', 'stress_solid': 'stress_normal_solid', 'stress_v2': 'stress_normal_solid',
            'stress_shear_sparse': 'stress_shear_sparse', 'velocity_fluid': 'velocity_fluid', 'velocity_v2': 'velocity_solid', 'velocity_solid': 'velocity_solid',
            'fused_fluid': 'fused_fluid'}
traffic = {'stress': 0.0, 'velocity': 0.0}
for k, d in agg.items():
    if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d and ('stress' in k or 'velocity' in k or 'fused' in k):
        f = sum(d['FETCH_SIZE']) / len(d['FETCH_SIZE']); w = sum(d['WRITE_SIZE']) / len(d['WRITE_SIZE'])
        b = (2 * f + w) * 1024     # gfx950: FETCH_SIZE counts half the read bytes
        if 'fused' not in k:
            traffic['stress' if 'stress' in k else 'velocity'] += b
        base = k.split('<')[0].replace('void ', '').strip()
        if base in CLASS_OF:
            traffic[CLASS_OF[base]] = traffic.get(CLASS_OF[base], 0.0) + b
if len(sys.argv) > 2:
    json.dump({sys.argv[2]: traffic}, open(root + '/traffic.json', 'w'))
for k, d in sorted(agg.items()):
    print('==', k)
    for c, v in sorted(d.items()):
        print('   %-28s n=%3d avg=%.6g' % (c, len(v), sum(v) / len(v)))
    if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
        f = sum(d['FETCH_SIZE']) / len(d['FETCH_SIZE']); w = sum(d['WRITE_SIZE']) / len(d['WRITE_SIZE'])
        print('   HBM bytes/launch (gfx950: FETCH_SIZE x2 KB + WRITE_SIZE KB): read %.3f GB write %.3f GB total %.3f GB'
              % (2 * f * 1024 / 1e9, w * 1024 / 1e9, (2 * f + w) * 1024 / 1e9))

"""Writes the round-6 entries of profiles/traffic.json (HBM bytes per time step and kernel class from the PMC passes, the kernel's average
duration under the profiler beside them) from what scripts/r6/profile_final.sh left in gpurun_out/r6_prof, and copies the summaries to profiles/r6."""
import csv, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
O = os.path.join(ROOT, 'gpurun_out', 'r6_prof')
P = os.path.join(ROOT, 'profiles', 'r6')
KEYS = {'c3': ('C3_512x512x512_variant0', 'c3_512'), 'c2': ('C2_512x512x512_variant0', 'c2medium_512'),
        'c4': ('C4_512x512x1024_variant0', 'c4_512x512x1024'), 'c5': ('C5_1024x1024x1024_variant0', 'c5_1024_cubed')}
CLASS_OF = {'stress_fluid': 'stress_fluid', 'stress_solid': 'stress_normal_solid', 'stress_shear_sparse': 'stress_shear_sparse',
            'velocity_fluid': 'velocity_fluid', 'velocity_solid': 'velocity_solid', 'fused_fluid': 'fused_fluid'}
tj = os.path.join(ROOT, 'profiles', 'traffic.json')
t = json.load(open(tj))
for w, (key, name) in KEYS.items():
    f = os.path.join(O, 'pmc_%s' % w, 'traffic.json')
    ks = os.path.join(O, 'kernel_stats_%s.csv' % w)
    if not (os.path.exists(f) and os.path.exists(ks)):
        print('skip', w); continue
    tr = json.load(open(f))[w]
    tot, steps = {}, 0
    for r in csv.DictReader(open(ks)):
        base = r['Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('<')[0].split('(')[0].strip()
        if base in CLASS_OF:
            c = CLASS_OF[base]
            tot[c] = tot.get(c, 0.0) + float(r['TotalDurationNs'])
            if base == 'velocity_fluid':
                steps += int(r['Calls'])
    avg = {c: round(v / steps / 1e3, 1) for c, v in tot.items()} if steps else {}
    shutil.copy(ks, os.path.join(P, 'kernel_stats_%s.csv' % name))
    shutil.copy(os.path.join(O, 'pmc_summary_%s.txt' % w), os.path.join(P, 'pmc_summary_%s.txt' % name))
    b = os.path.join(O, 'bench_%s_under_rocprof.json' % w)
    if os.path.exists(b) and os.path.getsize(b):
        shutil.copy(b, os.path.join(P, 'bench_%s_under_rocprof.json' % name))
    tr['_profile'] = {'pmc': 'profiles/r6/pmc_summary_%s.txt' % name, 'kernel_stats': 'profiles/r6/kernel_stats_%s.csv' % name, 'kernel_avg_us': avg,
                      'note': 'per class and time step (a class may take several launches per step); round 6, final binary (scripts/r6/profile_final.sh)'}
    if key in t and 'r5_' + key not in t:
        t['r5_' + key] = t[key]
    t[key] = tr
    print(key, {c: round(v / 1e9, 3) for c, v in tr.items() if not c.startswith('_')}, avg)
json.dump(t, open(tj, 'w'), indent=1)

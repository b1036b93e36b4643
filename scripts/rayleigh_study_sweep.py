#!/usr/bin/env python3
"""Runs Single-transducer water cases of the reference's Rayleigh-vs-FDTD study (tests/rayleigh_study.py) through the
drop-in on the GPU and prints this engine's metrics beside the workbook's.
  python scripts/rayleigh_study_sweep.py [--zadj 0 -10] [--every N] [--depth-mm 65] [--out gpurun_out/study.json]
BABELFDTD_HIP_LIB selects a variant build of the library (scheme experiments)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--zadj', type=float, nargs='*', default=[0.0, -10.0, 10.0])
    ap.add_argument('--tx', default='Single', choices=['Single', 'CTX_500', 'H317', 'REMOPD'])
    ap.add_argument('--every', type=int, default=1)
    ap.add_argument('--freq-khz', type=int, default=None, help='H317: only the cases of this frequency')
    ap.add_argument('--cases', type=int, nargs='*', default=None)
    ap.add_argument('--depth-mm', type=float, default=None)
    ap.add_argument('--gap-vox', type=float, default=1.0)
    ap.add_argument('--pml', type=int, default=None, help='absorbing-layer thickness (sensitivity experiments; the reference uses 12)')
    ap.add_argument('--out', default=None)
    args = ap.parse_args()
    from babelbrain_amd import PropagationModel, RayleighAndBHTE as R, _engine
    from tests import rayleigh_study as RS
    rows = [c for c in json.load(open(os.path.join(ROOT, 'tests', 'golden', 'rayleigh_study.json')))['cases'] if c['tx'] == args.tx]
    if args.tx not in ('H317', 'REMOPD'):
        rows = [r for r in rows if float(r['Description'].split('_')[1]) in args.zadj]
    else:
        rows = [r for r in rows if args.freq_khz is None or ('_%dkHz_' % args.freq_khz) in r['Description']]
    if args.cases is not None:
        rows = [r for r in rows if r['case'] in args.cases]
    rows = rows[::args.every]
    model = PropagationModel()
    dt_fn = lambda ml, f, h, c: _engine.stable_dt(ml, f, True, h, c)
    solver = lambda *a, **k: model.StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    depth = RS.DEPTH_TARGET if args.depth_mm is None else args.depth_mm * 1e-3
    out = []
    for r in rows:
        t = time.time()
        m = RS.run_case(r, solver, dt_fn, R.ForwardSimple, depth, args.gap_vox, args.pml)
        m['case'] = r['case']; m['Description'] = r['Description']; m['seconds'] = time.time() - t
        m['ref'] = {k: r[k] for k in ('Difference amplitude', 'L2', 'L Inf', 'Distance focal centroid', 'L Inf location')}
        out.append(m)
        print('%3d %-58s amp %+5.2f (%+5.2f)  L2 %5.2f (%5.2f)  Linf %5.2f (%5.2f)  cent %4.2f (%4.2f)  N %s nt %d ppp %d cfl %.3f  %.1fs'
              % (r['case'], r['Description'][:58], m['Difference amplitude'], r['Difference amplitude'], m['L2'], r['L2'], m['L Inf'], r['L Inf'],
                 m['Distance focal centroid'], r['Distance focal centroid'], m['N'], m['nt'], m['ppp'], m['cfl_water'], m['seconds']), flush=True)
        if args.tx in ('H317', 'REMOPD'):
            print('      Linf location %s (%s)' % (m['L Inf location'], r['L Inf location']), flush=True)
    a = np.array([[m['Difference amplitude'], m['ref']['Difference amplitude'], m['L2'], m['ref']['L2']] for m in out])
    print('median amp diff %.3f (workbook %.3f); median L2 %.3f (workbook %.3f); mean |amp - ref| %.3f; mean L2 ratio %.3f'
          % (np.median(a[:, 0]), np.median(a[:, 1]), np.median(a[:, 2]), np.median(a[:, 3]), np.mean(np.abs(a[:, 0] - a[:, 1])), np.mean(a[:, 2] / a[:, 3])))
    if args.out:
        json.dump({'lib': _engine.LIB_PATH, 'depth_target_m': depth, 'cases': out}, open(args.out, 'w'))


if __name__ == '__main__':
    main()

"""Round 6 hunt: quiet runs in Z-slabs. The random media of quiet_runs_hunt.py as a group of 2-4 slabs on device 0 (bfd_group_*), with
BFD_SKIP_ZERO_SLABS=1 against =0 and against ONE engine with every run working; tolerance zero on sensors, last / RMS / peak maps.
usage: quiet_slabs_hunt.py first_seed count"""
import os, sys, time
sys.path.insert(0, '.')
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from quiet_runs_hunt import big_case, run as run_single
from babelbrain_amd import _engine
from babelbrain_amd.PropagationModel import compact_sources


def run_group(a, k, nslab, mode):
    os.environ['BFD_SKIP_ZERO_SLABS'] = mode
    os.environ['BFD_SKIP_ZERO'] = '1'
    mm, ml, f, src, pulse, h, nt, sens = a
    g = _engine.Group([0] * nslab, *mm.shape, len(ml), h, k['DT'], f, nt, NDelta=k['NDelta'], typeSource=k['TypeSource'], sensorSub=k['sub'], sensorStart=0,
                      selMapsRMS=k['maps'], selMapsSensors=k['sens_maps'], selRMSorPeak=k['SelRMSorPeak'])
    g.set_materials(ml, k['QCorrection']); g.set_material_map(mm)
    if k['refl'] is not None:
        g.set_reflector(k['refl'])
    g.set_sources(*compact_sources(src, np.array([0.3]), np.array([0.7]), np.array([1.0])), pulse)
    g.set_sensor_map(sens)
    half = nt // 2
    g.run(half)
    g.sync()
    act = [0, 0]
    for r in range(g.size):
        a_, t_ = g.slab(r)[3].activity_counts() if hasattr(g.slab(r)[3], 'activity_counts') else (0, 0)
        act[0] += a_; act[1] += t_
    g.run(nt - half)
    out = {'sensors': g.sensors()}
    for n in k['maps']:
        out['last_' + n] = g.get_map(_engine.KIND_LAST, n)
        if k['SelRMSorPeak'] & 1: out['rms_' + n] = g.get_map(_engine.KIND_RMS, n)
        if k['SelRMSorPeak'] & 2: out['peak_' + n] = g.get_map(_engine.KIND_PEAK, n)
    g.close()
    return out, act


def main():
    first, count = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 40)
    bad, clear, t0 = [], [], time.time()
    for seed in range(first, first + count):
        try:
            a, k = big_case(seed)
            nslab = 2 + seed % 3
            while nslab > 1 and a[0].shape[2] // nslab < 24:
                nslab -= 1
            on, act = run_group(a, k, nslab, '1')
            off, _ = run_group(a, k, nslab, '0')
            one, _ = run_single(a, k, '0')
            diff = [n for n in on if not np.array_equal(on[n], off[n])] + ['single:' + n for n in on if not np.array_equal(on[n], one[n])]
            if diff:
                bad.append(seed); print('MISMATCH seed', seed, a[0].shape, nslab, 'slabs', diff[:6], flush=True)
            if act[1]:
                clear.append(1.0 - act[0] / act[1])
        except Exception as e:
            bad.append(seed); print('ERROR seed', seed, repr(e)[:300], flush=True)
    print('quiet runs in Z-slabs (2-4 slabs) against every run working and against one engine: %d seeds (%d..%d) in %.0f s, %d bad; '
          'slabs with a map: sub-tiles still clear half way through %.0f %% on average (%d seeds)'
          % (count, first, first + count - 1, time.time() - t0, len(bad), 100 * (np.mean(clear) if clear else 0), len(clear)))
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()

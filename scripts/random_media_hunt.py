"""One-off hunt: many more seeds of tests/test_random_media_gpu.py than the suite runs, held to exact equality with the oracle."""
import sys, numpy as np, time
sys.path.insert(0, '.')
from tests.test_random_media_gpu import random_case
from tests.util import compare_runs
from oracle import oracle as O
from babelbrain_amd import PropagationModel
bad = []
t0 = time.time()
# usage: random_media_hunt.py [first_seed count]  -> only the single-domain hunt over that range (default: every section with its fixed seeds)
RANGE = range(int(sys.argv[1]), int(sys.argv[1]) + int(sys.argv[2])) if len(sys.argv) > 2 else range(14, 214)
for seed in RANGE:
    try:
        a, k = random_case(seed)
        oh = PropagationModel().StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
        orf = O.StaggeredFDTD_3D_with_relaxation(*a, **k)
        w = compare_runs(oh, orf, 0.0, both=(k['SelRMSorPeak'] == 3))
    except AssertionError as e:
        bad.append((seed, str(e)[:200])); print('MISMATCH seed', seed, str(e)[:200], flush=True)
    except Exception as e:
        bad.append((seed, repr(e)[:200])); print('ERROR seed', seed, repr(e)[:300], flush=True)
print('%d seeds (%d..%d) in %.0f s, %d bad' % (len(RANGE), RANGE[0], RANGE[-1], time.time() - t0, len(bad)))
if len(sys.argv) > 2:
    sys.exit(1 if bad else 0)

# the same media cut into 2-4 slabs (both step orders)
import torch
from babelbrain_amd import slab
from babelbrain_amd._engine import HALO_STRESS, HALO_VELOCITY
from tests.test_slab_gpu import _exchange
bad = []
t0 = time.time()
for seed in range(300, 360):
    try:
        a, k = random_case(seed)
        world = 2 + seed % 3
        if a[0].shape[2] < 12 * world:
            world = 2
        split = seed % 2 == 0
        ref = PropagationModel().StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
        slabs, infos = zip(*[slab.create_hip_slab(a, k, r, world, 0, kernelVariant=0) for r in range(world)])
        for _ in range(ref[-1]['nt']):
            if split:
                for s in slabs: s.half_step_stress(1)
                _exchange(slabs, HALO_STRESS)
                for s in slabs: s.half_step_stress(2)
                for s in slabs: s.half_step_velocity(1)
                _exchange(slabs, HALO_VELOCITY)
                for s in slabs: s.half_step_velocity(2)
            else:
                _exchange(slabs, HALO_VELOCITY)
                for s in slabs: s.half_step_stress()
                _exchange(slabs, HALO_STRESS)
                for s in slabs: s.half_step_velocity()
        torch.cuda.synchronize()
        m = slab.merge_slab_outputs([slab.collect_slab_outputs(s.eng, k, i) for s, i in zip(slabs, infos)])
        ok = all(np.array_equal(m['Sensor'][n], ref[0][n]) for n in k['SelMapsSensorsList']) and all(np.array_equal(m['LastMap'][n], ref[1][n]) for n in ref[1])
        if k['SelRMSorPeak'] & 1: ok = ok and all(np.array_equal(m['RMS'][n], ref[2][n]) for n in ref[1])
        if k['SelRMSorPeak'] & 2: ok = ok and all(np.array_equal(m['Peak'][n], ref[-2][n]) for n in ref[1])
        for s in slabs: s.eng.close()
        if not ok:
            bad.append(seed); print('SLAB MISMATCH seed', seed, a[0].shape, world, split, flush=True)
    except Exception as e:
        bad.append(seed); print('SLAB ERROR seed', seed, repr(e)[:300], flush=True)
print('slabs: 60 seeds in %.0f s, %d bad' % (time.time() - t0, len(bad)))

# engine options that must not change a bit: streamed source table, graph replay, run length / order, fused variant, in-loop DFT
import os
def same(o1, o2, k):
    ok = all(np.array_equal(o1[0][n], o2[0][n]) for n in o1[0]) and all(np.array_equal(o1[1][n], o2[1][n]) for n in o1[1])
    return ok and all(np.array_equal(o1[2][n], o2[2][n]) for n in o1[2])
bad = []
t0 = time.time()
for seed in range(400, 460):
    try:
        a, k = random_case(seed)
        ref = PropagationModel().StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, ReturnSensorDFT=True, **k)
        trials = [('BFD_SOURCE_TILE', str(1 + seed % 37)), ('BFD_USE_GRAPH', '1'), ('BFD_ZRUN', '8' if seed % 2 else '32'), ('BFD_RUN_ORDER', str(seed % 2))]
        for name, val in trials:
            os.environ[name] = val
            out = PropagationModel().StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
            del os.environ[name]
            if not same(ref, out, k):
                bad.append((seed, name)); print('OPTION MISMATCH seed', seed, name, val, flush=True)
        if k['TypeSource'] < 2:
            out = PropagationModel(kernelVariant=4).StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
            if not same(ref, out, k):
                bad.append((seed, 'variant4')); print('OPTION MISMATCH seed', seed, 'variant 4', flush=True)
        out = PropagationModel().StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, ReturnSensorDFT=True, ReturnSensorSeries=False, **k)
        for n in ref[-1]['SensorDFT']:
            if not (np.array_equal(ref[-1]['SensorDFT'][n], out[-1]['SensorDFT'][n]) and np.array_equal(ref[-1]['SensorPeak'][n], out[-1]['SensorPeak'][n])):
                bad.append((seed, 'dft')); print('OPTION MISMATCH seed', seed, 'in-loop DFT', n, flush=True)
    except Exception as e:
        bad.append((seed, 'error')); print('OPTION ERROR seed', seed, repr(e)[:300], flush=True)
print('options: 60 seeds in %.0f s, %d bad' % (time.time() - t0, len(bad)))

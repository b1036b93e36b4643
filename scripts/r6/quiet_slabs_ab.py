"""Round 6: what quiet runs buy a call that is cut into Z-slabs (the one-process group path behind PropagationModel(devices=[...]); here both slabs on
device 0): C3 at 512^3, the first 3000 steps of a production call, with the slabs' interior runs allowed to return at entry (default) and not
(BFD_SKIP_ZERO_SLABS=0), and results compared."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from babelbrain_amd import harness as H, PropagationModel, _engine, RayleighAndBHTE
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
a, k, info = H.make_problem('C3', steps=steps, stable_dt_fn=lambda ml, f, h, c: _engine.stable_dt(ml, f, True, h, c), forward=RayleighAndBHTE.ForwardSimple, full_sensors=False)
out = {}
for mode in ('1', '0', '1', '0'):
    os.environ['BFD_SKIP_ZERO_SLABS'] = mode
    pm = PropagationModel(devices=[0, 0])
    t0 = time.time(); r = pm.StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k); wall = time.time() - t0
    print('BFD_SKIP_ZERO_SLABS=%s: two slabs, %d steps: step loop %.2f s, call %.2f s, %.1f Gvoxel-steps/s' % (mode, steps, pm.last_timing['total_ms'] / 1e3, wall, 512 ** 3 * steps / pm.last_timing['total_ms'] / 1e6), flush=True)
    out.setdefault(mode, r[2]['Pressure'])
print('results equal:', bool(np.array_equal(out['1'], out['0'])))

# plain-step throughput with and without graph replay (production-shaped calls: accumulation and sensors only in the last 2 periods)
for N in "128 128 128" "192 192 256" "256 256 256"; do for g in 1 0; do
if [ $g = 1 ]; then export BFD_USE_GRAPH=1; else unset BFD_USE_GRAPH; fi
python - $N <<'PY'
import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd())
from babelbrain_amd import harness as H, PropagationModel, _engine
N = tuple(int(x) for x in sys.argv[1:4])
dtfn = lambda ml, f, h, a: _engine.stable_dt(ml, f, True, h, a)
a, k, info = H.make_problem('C1', N=N, steps=1200, stable_dt_fn=dtfn)
pm = PropagationModel()
pm.StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
t = pm.last_timing
print('N', N, 'graph' if os.environ.get('BFD_USE_GRAPH') else 'direct', 'steps', info['nt'], 'device ms/step %.4f' % (t['total_ms'] / info['nt']), 'Gvox/s %.1f' % (t['voxel_steps'] / t['total_ms'] / 1e6))
PY
done; done

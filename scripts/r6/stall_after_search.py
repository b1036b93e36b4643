"""Round 6: does the multi-second stall that follows a placement search (profiles/r6/d2h_into_untouched_memory.txt) go away behind a long step loop?
One drop-in call at C3 with `steps` time steps; BFD_PLACEMENT_FORCE_WALK=n makes the search draw, hold and release n more candidates."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import importlib
from babelbrain_amd import harness as H, _engine, RayleighAndBHTE
PMmod = importlib.import_module('babelbrain_amd.PropagationModel')
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
a, k, info = H.make_problem('C3', steps=steps, stable_dt_fn=lambda ml, f, h, c: _engine.stable_dt(ml, f, True, h, c), forward=RayleighAndBHTE.ForwardSimple)
log = []
E = _engine.Engine
class Spy(E):
    def __init__(self, *aa, **kk):
        super().__init__(*aa, **kk)
        for n in ('run', 'sensors', 'get_map', 'timing_end', 'set_sensor_map', 'set_material_map', 'set_sources'):
            f = getattr(self, n)
            def w(*a2, _f=f, _n=n, **k2):
                t = time.perf_counter(); r = _f(*a2, **k2); log.append((_n, time.perf_counter() - t)); return r
            setattr(self, n, w)
PMmod.Engine = Spy
pm = PMmod.PropagationModel()
t0 = time.perf_counter(); out = pm.StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k); wall = time.perf_counter() - t0
tot = {}
for n, t in log: tot[n] = tot.get(n, 0.0) + t
print('%d steps, FORCE_WALK=%s, D2H_THREADS=%s: call %.2f s, step loop %.2f s; %s' % (steps, os.environ.get('BFD_PLACEMENT_FORCE_WALK'), os.environ.get('BFD_D2H_THREADS'), wall, pm.last_timing['total_ms'] / 1e3,
      ', '.join('%s %.2f' % (n, t) for n, t in sorted(tot.items(), key=lambda x: -x[1]))), flush=True)

"""Round 6: where the wall time of ONE drop-in call through the one-process group path goes (PropagationModel(devices=[0, 0]) on the 1-GPU box), C3 at 512^3."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import importlib
from babelbrain_amd import harness as H, _engine, RayleighAndBHTE
PMmod = importlib.import_module('babelbrain_amd.PropagationModel')
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
nslab = int(sys.argv[2]) if len(sys.argv) > 2 else 2
a, k, info = H.make_problem('C3', steps=steps, stable_dt_fn=lambda ml, f, h, c: _engine.stable_dt(ml, f, True, h, c), forward=RayleighAndBHTE.ForwardSimple)
log = []
G = _engine.Group
class Spy(G):
    def __init__(self, *aa, **kk):
        t = time.perf_counter(); super().__init__(*aa, **kk); log.append(('Group()', time.perf_counter() - t))
        for n in ('set_materials', 'set_material_map', 'set_sources', 'set_sensor_map', 'run', 'sensors', 'sensor_index', 'get_map', 'close', 'timing_end', 'prepare'):
            if hasattr(self, n):
                f = getattr(self, n)
                def w(*a2, _f=f, _n=n, **k2):
                    t = time.perf_counter(); r = _f(*a2, **k2); log.append((_n, time.perf_counter() - t)); return r
                setattr(self, n, w)
_engine.Group = Spy
cs = PMmod.compact_sources
def cs_spy(*aa, **kk):
    t = time.perf_counter(); r = cs(*aa, **kk); log.append(('compact_sources', time.perf_counter() - t)); return r
PMmod.compact_sources = cs_spy
pm = PMmod.PropagationModel(devices=[0] * nslab)
for rep in range(2):
    del log[:]
    t0 = time.perf_counter(); out = pm.StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k); wall = time.perf_counter() - t0
    tot = {}
    for n, t in log: tot[n] = tot.get(n, 0.0) + t
    print('call %d: %d slabs, %d steps, %d sensors: wall %.2f s, step loop %.2f s' % (rep, nslab, steps, out[0]['Pressure'].shape[0], wall, pm.last_timing['total_ms'] / 1e3))
    for n, t in sorted(tot.items(), key=lambda x: -x[1]): print('    %-18s %.3f s' % (n, t))
    print('    %-18s %.3f s' % ('(unaccounted)', wall - sum(tot.values())), flush=True)
    del out

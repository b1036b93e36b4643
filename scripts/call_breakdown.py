"""Where the wall time of one drop-in call goes (host preparation, uploads, step loop, downloads): C2 at 256^3 with its
2000 steps and C3 at 512^3 with 300 steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import importlib
from babelbrain_amd import harness as H, _engine, RayleighAndBHTE
PMmod = importlib.import_module('babelbrain_amd.PropagationModel')
PropagationModel = PMmod.PropagationModel

def timed(obj, name, log):
    f = getattr(obj, name)
    def w(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); log.append((name, time.perf_counter() - t)); return r
    setattr(obj, name, w)

for cfg, steps in (('C2', 2000), ('C3', 300)):
    a, k, info = H.make_problem(cfg, steps=steps, stable_dt_fn=lambda ml, f, h, c: _engine.stable_dt(ml, f, True, h, c), forward=RayleighAndBHTE.ForwardSimple)
    log = []
    E = _engine.Engine
    class Spy(E):
        def __init__(self, *aa, **kk):
            t = time.perf_counter(); super().__init__(*aa, **kk); log.append(('Engine()', time.perf_counter() - t))
            for n in ('set_materials', 'set_material_map', 'set_sources', 'set_sensor_map', 'run', 'sensors', 'sensor_index', 'get_map', 'close', 'timing_end'):
                timed(self, n, log)
    PMmod.Engine = Spy
    cs = PMmod.compact_sources
    def cs_spy(*aa, **kk):
        t = time.perf_counter(); r = cs(*aa, **kk); log.append(('compact_sources', time.perf_counter() - t)); return r
    PMmod.compact_sources = cs_spy
    pm = PropagationModel()
    pm.StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)       # first call of the process: library load, code object upload
    del log[:]
    t0 = time.perf_counter(); out = pm.StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k); wall = time.perf_counter() - t0
    PMmod.Engine = E; PMmod.compact_sources = cs
    tot = {}
    for n, t in log: tot[n] = tot.get(n, 0.0) + t
    print('%s %s, %d steps: call %.3f s, device step loop %.3f s' % (cfg, a[0].shape, info['nt'], wall, pm.last_timing['total_ms'] / 1e3))
    for n, t in sorted(tot.items(), key=lambda x: -x[1]): print('    %-18s %.3f s' % (n, t))
    print('    %-18s %.3f s' % ('(unaccounted)', wall - sum(tot.values())))

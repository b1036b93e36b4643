#!/usr/bin/env python3
"""What bfd_prepare's placement costs in a fresh process: engines of 5 M ... 134 M voxels, one after the other."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from babelbrain_amd import _engine, harness as H
dt_fn = lambda ml, f, h, c: _engine.stable_dt(ml, f, True, h, c)
for cfg, N in (('C2', (176, 176, 176)), ('C3', (256, 256, 256)), ('C3', (512, 512, 512)), ('C2', (512, 512, 512)), ('C2', (176, 176, 176))):
    a, k, info = H.make_problem(cfg, N=N, steps=10, stable_dt_fn=dt_fn, full_sensors=False)
    e = _engine.Engine(*N, len(a[1]), a[5], k['DT'], a[2], 10, sensorSub=k['SensorSubSampling'], sensorStart=k['SensorStart'])
    e.set_materials(a[1], k.get('QCorrection', 1.0)); e.set_material_map(a[0], 0, 0)
    t = time.time(); e.prepare(); e.sync(); dt = time.time() - t
    print(cfg, N, 'prepare %.3f s:' % dt, e.placement_note()[:400], flush=True)
    e.close()

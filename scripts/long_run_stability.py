import sys, os, numpy as np
sys.path.insert(0, '/root/repo')
from babelbrain_amd import harness as H, _engine
from babelbrain_amd.PropagationModel import compact_sources
dtfn = lambda ml, f, h, a: _engine.stable_dt(ml, f, True, h, a)
for cfg, N in (('C2', (128, 128, 160)), ('C3', (128, 128, 160)), ('C1', (96, 96, 128))):
    nt = 12000
    a, k, info = H.make_problem(cfg, N=N, steps=nt, stable_dt_fn=dtfn)
    mm, ml, f, smap, pulse, h, T, sensor = a
    eng = _engine.Engine(N[0], N[1], N[2], len(ml), h, k['DT'], f, nt, sensorSub=k['SensorSubSampling'], sensorStart=k['SensorStart'],
                         selMapsRMS=['Pressure'], selMapsSensors=['Pressure'], kernelVariant=0)
    eng.set_materials(ml, k['QCorrection']); eng.set_material_map(mm, 0, 0)
    eng.set_sources(*compact_sources(smap, k['Ox'], k['Oy'], k['Oz']), pulse); eng.set_sensor_map(sensor)
    out = []
    for chunk in range(12):
        eng.run(1000 if chunk < 11 else nt - 11000)
        v = [np.abs(eng.get_field(n)).max() for n in ('Szz', 'Sxy', 'Vz')]
        out.append(v)
        print(cfg, 'step', eng.step, 'max |Szz| %.4g |Sxy| %.4g |Vz| %.4g' % tuple(v), flush=True)
    eng.close()

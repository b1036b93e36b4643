"""Round 6: what quiet runs buy over a production-length call at C3 (512^3, nt from the caller's time plan): ms per step and the share of active
sub-tiles in blocks of 500 steps, with the runs ahead of the front returning at entry (default) and with every run working (BFD_SKIP_ZERO=0)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from babelbrain_amd import harness as H, _engine, RayleighAndBHTE
from babelbrain_amd.PropagationModel import compact_sources
cfg = sys.argv[1] if len(sys.argv) > 1 else 'C3'
size = tuple(int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else None
a, k, info = H.make_problem(cfg, N=size, stable_dt_fn=lambda ml, f, h, c: _engine.stable_dt(ml, f, True, h, c), forward=RayleighAndBHTE.ForwardSimple, full_sensors=False)
mm, ml, f, smap, pulse, h, T, sensor = a
nt = info['nt']
print('%s %s nt=%d ppp=%d' % (cfg, mm.shape, nt, info['ppp']), flush=True)
src = compact_sources(smap, k['Ox'], k['Oy'], k['Oz'])
for mode in ('1', '0', '1', '0'):
    os.environ['BFD_SKIP_ZERO'] = mode
    eng = _engine.Engine(*mm.shape, len(ml), h, k['DT'], f, nt, sensorSub=k['SensorSubSampling'], sensorStart=k['SensorStart'], selMapsRMS=['Pressure'],
                         selMapsSensors=['Pressure'], selRMSorPeak=1)
    eng.set_materials(ml, k['QCorrection']); eng.set_material_map(mm, 0, 0); eng.set_sources(*src, pulse); eng.set_sensor_map(sensor)
    eng.run(1); eng.reset()
    rows, tot, done = [], 0.0, 0
    while done < nt:
        n = min(500, nt - done)
        eng.timing_begin(False); eng.run(n); tm = eng.timing_end()
        done += n; tot += tm['total_ms']
        act = eng.activity_counts()
        rows.append('%5d: %.3f ms/step%s' % (done, tm['total_ms'] / n, (' active %.2f' % (act[0] / act[1])) if act[1] else ''))
    print('BFD_SKIP_ZERO=%s: %.2f s for %d steps = %.1f Gvoxel-steps/s' % (mode, tot / 1e3, nt, float(np.prod(mm.shape)) * nt / tot / 1e6))
    print('   ' + '\n   '.join(rows), flush=True)
    eng.close()

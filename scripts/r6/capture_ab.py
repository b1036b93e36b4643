"""Round 6: one sensor capture at C3 (116 M sensors in a dense box) -- the division-free box kernel against the list kernel's arithmetic on the same box
(BFD_SENSOR_BOX_KERNEL=0) and against the index list (BFD_SENSOR_BOX=0); each in a child process, timed with events over 20 captures."""
import os, sys, subprocess
if len(sys.argv) > 1:
    sys.path.insert(0, '.')
    import numpy as np
    from babelbrain_amd import harness as H, _engine, RayleighAndBHTE
    from babelbrain_amd.PropagationModel import compact_sources
    a, k, info = H.make_problem('C3', steps=60, stable_dt_fn=lambda ml, f, h, c: _engine.stable_dt(ml, f, True, h, c), forward=RayleighAndBHTE.ForwardSimple)
    mm, ml, f, smap, pulse, h, T, sens = a
    src = compact_sources(smap, k['Ox'], k['Oy'], k['Oz'])
    def run(sub):
        os.environ['BFD_SKIP_ZERO'] = '0'
        eng = _engine.Engine(*mm.shape, len(ml), h, k['DT'], f, 60, sensorSub=sub, sensorStart=0, selMapsRMS=['Pressure'], selMapsSensors=['Pressure'], selRMSorPeak=1)
        eng.set_materials(ml, k['QCorrection']); eng.set_material_map(mm, 0, 0); eng.set_sources(*src, pulse); eng.set_sensor_map(sens)
        eng.run(8 if sub == 1 else 8)
        eng.timing_begin(False); eng.run(12); tm = eng.timing_end()
        out = eng.sensors() if sub == 1 else None
        eng.close()
        return tm['total_ms'] / 12, out
    t1, o1 = run(1)          # a capture in every step
    t0, _ = run(1000)        # none in the window
    print('%s: step with a capture %.3f ms, without %.3f ms -> capture %.3f ms; checksum %.9e' % (sys.argv[1], t1, t0, t1 - t0, float(np.abs(o1).sum(dtype=np.float64))), flush=True)
else:
    for name, env in (('box kernel', {}), ('list arithmetic on the box', {'BFD_SENSOR_BOX_KERNEL': '0'}), ('index list', {'BFD_SENSOR_BOX': '0'})):
        subprocess.run([sys.executable, __file__, name], env=dict(os.environ, **env))

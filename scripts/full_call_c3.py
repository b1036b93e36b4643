"""One production-shaped drop-in call at C3 (512^3, nt from the caller's time plan, full sensor volume):
wall time of PropagationModel.StaggeredFDTD_3D_with_relaxation including upload, layout conversion, the step loop
and the download of the sensor block and maps -- the PCIe-inclusive rate quoted in DESIGN.md."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from babelbrain_amd import harness as H, PropagationModel, _engine, RayleighAndBHTE
t0 = time.time()
a, k, info = H.make_problem('C3', stable_dt_fn=lambda ml, f, h, c: _engine.stable_dt(ml, f, True, h, c), forward=RayleighAndBHTE.ForwardSimple)
t1 = time.time()
print('inputs built in %.1f s: nt=%d ppp=%d sub=%d start=%d PulseSource %.1f GB' % (t1 - t0, info['nt'], info['ppp'], k['SensorSubSampling'], k['SensorStart'], a[4].nbytes / 1e9), flush=True)
pm = PropagationModel(keepPlacementCache=True)      # the calls of one RUN_SIMULATION, back to back
N = 512 ** 3
for series in (True, False):      # the reference's return values; then without the sensor series (DFT accumulated in the loop)
    t2 = time.time()
    out = pm.StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, ReturnSensorDFT=True, ReturnSensorSeries=series, **k)
    t3 = time.time()
    tm = pm.last_timing
    print('ReturnSensorSeries=%s: call wall %.2f s; step loop (device) %.2f s; voxel-steps %.3e; device memory %.1f GB'
          % (series, t3 - t2, tm['total_ms'] / 1e3, N * info['nt'], out[-1]['device_bytes'] / 1e9))
    print('   device-only %.0f Mvoxel-steps/s; PCIe-inclusive (whole call) %.0f Mvoxel-steps/s'
          % (N * info['nt'] / tm['total_ms'] / 1e3, N * info['nt'] / (t3 - t2) / 1e6))
    print('   placement:', out[-1].get('placement'))
    print('   timing:', {k: (round(v, 2) if isinstance(v, float) else v) for k, v in tm.items()})
    if series:
        print('   sensor block', out[0]['Pressure'].shape, '%.1f GB' % (out[0]['Pressure'].nbytes / 1e9), 'RMS max', float(out[2]['Pressure'].max()))
        ref = out[-1]['SensorDFT']['Pressure']
    else:
        print('   in-loop DFT equals the DFT of the series:', bool(np.array_equal(ref, out[-1]['SensorDFT']['Pressure'])))
    del out

"""Does the ±8 % process-to-process spread of velocity_fluid come from where the arrays land in memory? One process builds
the C3 engine several times, with a throw-away allocation of varying size in between, and times the kernels each time.
BFD_PLACEMENT_TRIALS=0 shows the raw placements (0.89 or 1.0 ms for the same kernel on the same data); the default lets
every engine choose among 1 + 3 sets of arrays (BFD_PLACEMENT_VERBOSE=1 prints the candidates)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from babelbrain_amd import harness as H, _engine, RayleighAndBHTE
from babelbrain_amd.PropagationModel import compact_sources, n_steps

a, k, info = H.make_problem('C3', steps=60, stable_dt_fn=lambda ml, f, h, c: _engine.stable_dt(ml, f, True, h, c), forward=RayleighAndBHTE.ForwardSimple,
                            accumulate_all_steps=True)
mm, ml, f, smap, pulse, h, T, sens = a
N1, N2, N3 = mm.shape
lin, row, wx, wy, wz = compact_sources(smap, k['Ox'], k['Oy'], k['Oz'])
keep = []
for trial in range(int(os.environ.get('PROBE_ENGINES', '8'))):
    eng = _engine.Engine(N1, N2, N3, len(ml), h, k['DT'], f, info['nt'], NDelta=k['NDelta'], reflectionLimit=k['ReflectionLimit'], typeSource=0,
                         sensorSub=k['SensorSubSampling'], sensorStart=k['SensorStart'], selRMSorPeak=1, selMapsRMS=['Pressure'],
                         selMapsSensors=['Pressure'], qfactorCorrection=True, device=0, rmsFirstStep=int(os.environ.get('PROBE_RMS_FIRST_STEP', '1')))
    eng.set_materials(ml, k['QCorrection']); eng.set_material_map(mm, 0, 0)
    eng.set_sources(lin, row, wx, wy, wz, pulse); eng.set_sensor_map(sens)
    eng.run(10); torch.cuda.synchronize()
    eng.timing_begin(2); eng.run(40); eng.timing_end(); kt = eng.timing_kernels()
    print('engine %d: ' % trial + '  '.join('%s %.3f ms' % (c, ms / max(n, 1)) for c, (ms, n) in kt.items() if n), flush=True)
    eng.close()
    # perturb the allocator: keep a block of odd size alive across the next engine
    keep.append(torch.empty(int((0.3 + float(os.environ.get('PROBE_STEP', '0.37')) * trial) * 2 ** 30), dtype=torch.uint8, device='cuda'))
    if len(keep) > 2: keep.pop(0)

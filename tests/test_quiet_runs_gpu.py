"""Quiet runs (bfd_dev::act, ABI 7): in a production call of a whole domain the tile runs ahead of the wave front -- every field there is exactly
zero -- return at entry. The results must not know: every test here compares bit for bit with the same engine under BFD_SKIP_ZERO=0 (every run
works in every half-step), on grids several tiles wide so that runs really are skipped, and checks through bfd_activity_counts that they were."""
import numpy as np
import pytest

from babelbrain_amd import _engine, harness as H
from babelbrain_amd.PropagationModel import compact_sources
from tests.util import compare_runs, oracle_dt

pytestmark = pytest.mark.gpu


def _hip_dt(ml, f, h, c):
    return _engine.stable_dt(ml, f, True, h, c)


def _engine_for(a, k, nt, maps=('Pressure',), **kw):
    mm, ml, f, smap, pulse, h, T, sensor = a
    eng = _engine.Engine(*mm.shape, len(ml), h, k['DT'], f, nt, NDelta=k['NDelta'], typeSource=k['TypeSource'], sensorSub=k['SensorSubSampling'],
                         sensorStart=k['SensorStart'], selMapsRMS=list(maps), selMapsSensors=['Pressure'], selRMSorPeak=1, **kw)
    eng.set_materials(ml, k['QCorrection'])
    eng.set_material_map(mm, 0, 0)
    if k.get('ReflectorMask') is not None:
        eng.set_reflector(k['ReflectorMask'])
    eng.set_sources(*compact_sources(smap, k['Ox'], k['Oy'], k['Oz']), pulse)
    eng.set_sensor_map(sensor)
    return eng


def _staged(a, k, stages, skip, monkeypatch, again_at=None, **kw):
    """Runs the stages one after the other; after each: all 15 state arrays, the RMS map, the sensors and the activity counts."""
    monkeypatch.setenv('BFD_SKIP_ZERO', '1' if skip else '0')
    nt = sum(stages)
    eng = _engine_for(a, k, nt, **kw)
    out = []
    for q, n in enumerate(stages):
        if again_at is not None and q == again_at:         # inputs set again in the middle of a run: from here on every sub-tile counts as active
            mm, ml, f, smap, pulse, h, T, sensor = a
            eng.set_sources(*compact_sources(smap, k['Ox'], k['Oy'], k['Oz']), pulse)
        eng.run(n)
        o = {n_: eng.get_field(n_).copy() for n_ in _engine.FIELD_NAMES}
        o['rms'] = eng.get_map(_engine.KIND_RMS, 'Pressure').copy()
        o['sensors'] = eng.sensors().copy()
        o['activity'] = eng.activity_counts()
        out.append(o)
    eng.reset()
    eng.run(stages[0])                                     # after a reset the map starts over
    o = {n_: eng.get_field(n_).copy() for n_ in _engine.FIELD_NAMES}
    o['activity'] = eng.activity_counts()
    out.append(o)
    eng.close()
    return out


def _same(on, off):
    for q, (x, y) in enumerate(zip(on, off)):
        for n in x:
            if n != 'activity':
                assert np.array_equal(x[n], y[n]), (q, n)


@pytest.mark.parametrize('config,N', [('C1', (192, 96, 160)), ('C2', (192, 96, 160)), ('C3', (160, 128, 144))])
def test_quiet_runs_do_not_change_a_bit(config, N, monkeypatch):
    """Water, skull with shear (compact solid state, the sparse kernel beside the marching ones) and the CT medium: stages of 60 + 200 + 400 steps,
    then a reset. While the front is on its way some sub-tiles are still clear (runs were skipped), later all of the interior is active."""
    a, k, info = H.make_problem(config, N=N, steps=660, stable_dt_fn=_hip_dt)
    on = _staged(a, k, (60, 200, 400), True, monkeypatch)
    off = _staged(a, k, (60, 200, 400), False, monkeypatch)
    _same(on, off)
    assert all(o['activity'] == (0, 0) for o in off)
    act = [o['activity'] for o in on]
    total = act[0][1]
    assert total == -(-N[0] // 64) * -(-N[1] // 8) * -(-N[2] // 8)
    assert 0 < act[0][0] < act[1][0] < act[2][0] <= total and act[0][0] < 0.5 * total, act
    assert act[3] == act[0]                                # the same stage after a reset marks the same sub-tiles
    assert on[2]['rms'].max() > 0 and np.abs(on[0]['Vz']).max() > 0


def test_quiet_runs_with_a_stress_source_a_reflector_and_every_map(monkeypatch):
    """The back-propagation call's shape (a stress point source, TypeSource 2), a reflector mask, all map selections accumulated outside the kernels."""
    from tests.util import ALL_MAPS
    a, k, info = H.make_problem('C2', N=(192, 96, 160), steps=420, stable_dt_fn=_hip_dt)
    mm, ml, f, smap, pulse, h, T, sensor = a
    smap2, pulse2 = H.punctual_source_map(*mm.shape, (120, 47, 100)), H.punctual_source(f, k['DT'], T, ramp_length=1)
    refl = np.zeros(mm.shape, np.uint32); refl[60:70, 30:60, 60:64] = 1
    k2 = dict(k, TypeSource=2, ReflectorMask=refl, Ox=np.array([1.0]), Oy=np.array([1.0]), Oz=np.array([1.0]))
    a2 = (mm, ml, f, smap2.astype(np.uint32), pulse2, h, T, sensor)
    on = _staged(a2, k2, (80, 340), True, monkeypatch, maps=ALL_MAPS)
    off = _staged(a2, k2, (80, 340), False, monkeypatch, maps=ALL_MAPS)
    _same(on, off)
    assert 0 < on[0]['activity'][0] < on[1]['activity'][0] and np.abs(on[1]['Sxy']).max() > 0


def test_inputs_set_again_in_the_middle_of_a_run(monkeypatch):
    a, k, info = H.make_problem('C2', N=(192, 96, 160), steps=400, stable_dt_fn=_hip_dt)
    on = _staged(a, k, (100, 300), True, monkeypatch, again_at=1)
    off = _staged(a, k, (100, 300), False, monkeypatch, again_at=1)
    _same(on, off)
    assert on[0]['activity'][0] < on[0]['activity'][1] and on[1]['activity'][0] == on[1]['activity'][1]


def test_drop_in_call_with_quiet_runs_against_the_oracle(monkeypatch):
    """The drop-in call (which is a production call: quiet runs on) against the oracle, and against itself with every run working, on a grid the front
    has not crossed when the run ends."""
    from babelbrain_amd import PropagationModel
    from oracle import oracle as O
    a, k, info = H.make_problem('C2', N=(192, 96, 160), steps=300, stable_dt_fn=oracle_dt)
    monkeypatch.setenv('BFD_SKIP_ZERO', '1')
    out_on = PropagationModel().StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    monkeypatch.setenv('BFD_SKIP_ZERO', '0')
    out_off = PropagationModel().StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    assert compare_runs(out_on, out_off, tol=0.0) == 0.0
    out_o = O.StaggeredFDTD_3D_with_relaxation(*a, **k)
    compare_runs(out_on, out_o, tol=1e-5)
    assert out_o[1]['Pressure'][:, :, -40:].max() == 0 and out_o[1]['Pressure'].max() > 0      # the far end is still untouched


def test_bench_windows_work_on_every_run(monkeypatch):
    """rmsFirstStep > 0 (bench.py's timed windows) keeps every run working: no map at all."""
    monkeypatch.setenv('BFD_SKIP_ZERO', '1')
    a, k, info = H.make_problem('C1', N=(128, 64, 96), steps=40, stable_dt_fn=_hip_dt)
    eng = _engine_for(a, k, 40, rmsFirstStep=1)
    eng.run(40)
    assert eng.activity_counts() == (0, 0)
    eng.close()


@pytest.mark.parametrize('config,split', [('C2', True), ('C2', False), ('C1', True)])
def test_quiet_runs_in_z_slabs(config, split, monkeypatch):
    """A production call cut into Z-slabs (the one-process group path and torchrun's slab.py drive the same engines): every slab keeps its own map, the
    runs of its first and last sub-tile -- the ones a neighbour's planes reach -- always work and wake the rest when the wave comes in. Three slabs with
    the halo planes exchanged by device copies, both step orders, against the single domain with every run working: bit for bit; and the slabs the wave has
    not reached yet show clear sub-tiles."""
    import torch
    from babelbrain_amd import PropagationModel, slab
    from babelbrain_amd._engine import HALO_STRESS, HALO_VELOCITY
    from tests.test_slab_gpu import _exchange
    N, steps, world = (128, 64, 288), 420, 3
    a, k, info = H.make_problem(config, N=N, steps=steps, stable_dt_fn=_hip_dt)
    monkeypatch.setenv('BFD_SKIP_ZERO', '0')
    ref = PropagationModel().StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    monkeypatch.setenv('BFD_SKIP_ZERO', '1')
    slabs, infos = zip(*[slab.create_hip_slab(a, k, r, world, 0, kernelVariant=0) for r in range(world)])
    seen = []
    for n in range(steps):
        if split:
            for s in slabs: s.half_step_stress(1)
            _exchange(slabs, HALO_STRESS)
            for s in slabs: s.half_step_stress(2)
            for s in slabs: s.half_step_velocity(1)
            _exchange(slabs, HALO_VELOCITY)
            for s in slabs: s.half_step_velocity(2)
        else:
            _exchange(slabs, HALO_VELOCITY)
            for s in slabs: s.half_step_stress()
            _exchange(slabs, HALO_STRESS)
            for s in slabs: s.half_step_velocity()
        if n in (60, steps - 1):
            torch.cuda.synchronize()
            seen.append([s.eng.activity_counts() for s in slabs])
    torch.cuda.synchronize()
    m = slab.merge_slab_outputs([slab.collect_slab_outputs(s.eng, k, i) for s, i in zip(slabs, infos)])
    for s in slabs: s.close()
    assert np.array_equal(m['RMS']['Pressure'], ref[2]['Pressure']) and np.array_equal(m['LastMap']['Pressure'], ref[1]['Pressure'])
    assert np.array_equal(m['Sensor']['Pressure'], ref[0]['Pressure'])
    early, late = seen
    assert all(t > 0 for _, t in early) and early[0][0] > 0 and early[2][0] == 0, early       # the wave starts in slab 0; slab 2 is still asleep after 60 steps
    assert late[0][0] > early[0][0] and late[1][0] > 0, (early, late)
    assert ref[1]['Pressure'][:, :, 96:192].max() > 0                                              # it did cross the first interface


@pytest.mark.timeout(900)
def test_production_length_call_at_c3_with_and_without_quiet_runs(monkeypatch):
    """The metric's own configuration over a whole production call (512^3, nt = 6760 from the caller's time plan): Pressure RMS, last map and
    the sensor lines of a call whose runs ahead of the front return at entry equal those of a call in which every run works, bit for bit.
    (What small grids cannot show: 2 x 10^8 workgroups take the decision while neighbours set their bytes in the same launch.)"""
    psutil = pytest.importorskip('psutil')
    if psutil.virtual_memory().available < 40 * 2 ** 30:
        pytest.skip('less than 40 GB of host memory for the 12.9 GB source table and the outputs of two calls')
    from babelbrain_amd import PropagationModel, RayleighAndBHTE
    a, k, info = H.make_problem('C3', stable_dt_fn=_hip_dt, forward=RayleighAndBHTE.ForwardSimple, full_sensors=False)
    assert a[0].shape == (512, 512, 512) and info['nt'] > 6000
    outs = []
    for mode in ('1', '0'):
        monkeypatch.setenv('BFD_SKIP_ZERO', mode)
        outs.append(PropagationModel().StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k))
    assert compare_runs(outs[0], outs[1], tol=0.0) == 0.0
    assert outs[0][2]['Pressure'].max() > 0
    t_on, t_off = outs[0][-1]['timing']['total_ms'], outs[1][-1]['timing']['total_ms']
    # (no assertion on the times: the first call of a process also pays the placement search inside its step-loop timer; scripts/r6/quiet_profile.py
    # and scripts/full_call_c3.py measure what the quiet runs save)
    print('C3 production call, step loop: %.2f s with quiet runs, %.2f s with every run working' % (t_on / 1e3, t_off / 1e3))

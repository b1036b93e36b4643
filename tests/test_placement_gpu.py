"""The placement of the per-voxel arrays by memory region (bfd_prepare) must never cost a caller the
device: its search for a buffer in another region holds throw-away allocations, so it is switched off on a device that carries
other allocations, bounded on a device of its own (192 GiB, at most two thirds of the free memory, 48 GiB always left; any explicit limit), and it
gives up quietly -- bfd_prepare succeeds wherever it would with the placement off, and the results do not depend on it."""
import subprocess
import sys
import threading
import time

import numpy as np
import pytest

from babelbrain_amd import _engine, harness as H
from babelbrain_amd.PropagationModel import compact_sources

pytestmark = pytest.mark.gpu


def _hip_dt(ml, f, h, c):
    return _engine.stable_dt(ml, f, True, h, c)


def _engine_for(a, k, info, **kw):
    mm, ml, f, smap, pulse, h, T, sensor = a
    N1, N2, N3 = mm.shape
    eng = _engine.Engine(N1, N2, N3, len(ml), h, k['DT'], f, info['nt'], sensorSub=k['SensorSubSampling'], sensorStart=k['SensorStart'],
                         selMapsRMS=['Pressure'], selMapsSensors=['Pressure'], rmsFirstStep=1, **kw)
    eng.set_materials(ml, k['QCorrection'])
    eng.set_material_map(mm, 0, 0)
    eng.set_sources(*compact_sources(smap, k['Ox'], k['Oy'], k['Oz']), pulse)
    eng.set_sensor_map(sensor)
    return eng


HOLDER = r'''
import sys, time, torch
gib = int(sys.argv[1])
held = [torch.empty(4 << 30, dtype=torch.uint8, device='cuda') for _ in range(gib // 4)]
torch.cuda.synchronize()
print('HOLDING', flush=True)
sys.stdin.readline()
'''


@pytest.mark.timeout(600)
def test_placement_under_memory_pressure():
    """Another process holds ~200 GiB of the device; a C3-size engine (512^3, 12 GB) is prepared beside it: it must succeed, say
    that it did not search (device shared), and give the results of a run with the placement off."""
    lib = _engine.load_library()
    a, k, info = H.make_problem('C3', steps=140, stable_dt_fn=_hip_dt, full_sensors=False)
    ref = _engine_for(a, k, info)
    ref.set_placement(0)
    ref.run(140)
    want = ref.get_map(_engine.KIND_RMS, 'Pressure')
    assert ref.placement_note() == 'off'
    ref.close()
    holder = subprocess.Popen([sys.executable, '-c', HOLDER, '200'], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
    try:
        line = holder.stdout.readline()
        assert 'HOLDING' in line, 'the holder process could not take 200 GiB: %r' % line
        eng = _engine_for(a, k, info)
        eng.prepare()                                   # must not fail, must not take the rest of the device
        note = eng.placement_note()
        print('placement beside a 200 GiB holder:', note)
        assert 'device shared' in note and 'no search beyond the own buffers' in note
        eng.run(140)
        got = eng.get_map(_engine.KIND_RMS, 'Pressure')
        eng.close()
        assert np.array_equal(got, want) and want.max() > 0
    finally:
        try:
            holder.stdin.write('\n'); holder.stdin.flush()
        except Exception:
            pass
        holder.wait(60)
    assert lib.bfd_device_count() > 0


@pytest.mark.timeout(600)
def test_explicit_limit_and_default_bound():
    """bfd_set_placement: an explicit limit of 0 bytes exchanges buffers only, one of 64 GiB holds no more than that; the default on
    a free device may search but holds at most 192 GiB at a time (the note reports what was released)."""
    a, k, info = H.make_problem('C3', steps=140, stable_dt_fn=_hip_dt, full_sensors=False)
    outs = []
    for limit in (0, 64 << 30, -1):
        eng = _engine_for(a, k, info)
        eng.set_placement(1, limit)
        eng.prepare()
        note = eng.placement_note()
        print('limit', limit, ':', note)
        assert note.startswith('arrays placed by memory region')
        if limit == 0:
            assert ' 0 fresh' in note and '0 candidates' in note
        else:
            gib = float(note.split('spacers (')[1].split(' GiB')[0])
            assert gib <= (64.0 if limit > 0 else 192.0) + 1e-6
        with pytest.raises(_engine.EngineError):
            eng.set_placement(0)                        # too late: the arrays are placed
        eng.run(140)
        outs.append(eng.get_map(_engine.KIND_RMS, 'Pressure'))
        eng.close()
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2]) and outs[0].max() > 0


@pytest.mark.timeout(600)
def test_engines_prepared_at_the_same_time_on_one_device():
    """Four slab engines of one 512 x 512 x 512 domain prepare at the same time on one device from four host threads (what several
    solver calls of one process, or a group's slabs, amount to): every prepare succeeds, none searches once the others' memory is
    there, and the group call over four slabs of that device still equals the single-device call."""
    from babelbrain_amd import PropagationModel
    a, k, info = H.make_problem('C3', steps=140, stable_dt_fn=_hip_dt, full_sensors=False)
    engines = [_engine_for(a, k, info) for _ in range(4)]
    errs, notes = [], [None] * 4

    def work(q):
        try:
            engines[q].prepare()
            notes[q] = engines[q].placement_note()
        except Exception as e:
            errs.append(repr(e))
    th = [threading.Thread(target=work, args=(q,)) for q in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for e in engines:
        e.close()
    assert not errs, errs
    assert all(n and ('device shared' in n or 'arrays placed' in n) for n in notes), notes
    one = PropagationModel(device=0).StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    many = PropagationModel(devices=[0, 0, 0, 0]).StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    assert np.array_equal(one[2]['Pressure'], many[2]['Pressure']) and one[2]['Pressure'].max() > 0
    assert len(many[-1]['placement']) == 4 and isinstance(one[-1]['placement'], str)


def test_placement_buffers_kept_between_engines_and_released(monkeypatch):
    """Buffers a placement search found in another region are kept when their engine is destroyed and offered to the next engine of the
    process (the solver calls of one RUN_SIMULATION pay a search once); bfd_placement_cache_release frees them. Forced on a small grid
    (short search): whatever the search finds, a second engine built the same way gives the same results, the cache never holds more
    than its bound, and after a release it is empty."""
    monkeypatch.setenv('BFD_PLACEMENT_MIN_VOXELS', '0')
    monkeypatch.setenv('BFD_PLACEMENT_SEARCH_MB', '64')
    monkeypatch.setenv('BABELFDTD_PLACEMENT_CACHE_GIB', '1')
    _engine.placement_cache_release()
    a, k, info = H.make_problem('C2', N=(96, 80, 72), steps=60, stable_dt_fn=_hip_dt, full_sensors=False)
    outs, notes = [], []
    for _ in range(3):
        eng = _engine_for(a, k, info)
        eng.run(60)
        outs.append(eng.get_map(_engine.KIND_RMS, 'Pressure'))
        notes.append(eng.placement_note())
        eng.close()
    print(notes)
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2]) and outs[0].max() > 0
    freed = _engine.placement_cache_release()
    assert 0 <= freed <= 1 << 30
    assert _engine.placement_cache_release() == 0


def test_drop_in_call_leaves_no_idle_buffers_by_default(monkeypatch):
    """A solver call through the drop-in gives the placement cache back when it returns (a process that goes on to something else on the device
    finds nothing idle of this library there); keepPlacementCache=True keeps it for the next call, within the cache's bound. A new engine of
    another array size evicts what the cache holds for the old size when it is created."""
    monkeypatch.setenv('BFD_PLACEMENT_MIN_VOXELS', '0')
    monkeypatch.setenv('BFD_PLACEMENT_SEARCH_MB', '64')
    monkeypatch.setenv('BABELFDTD_PLACEMENT_CACHE_GIB', '1')
    from babelbrain_amd import PropagationModel
    _engine.placement_cache_release()
    a, k, info = H.make_problem('C2', N=(96, 80, 72), steps=40, stable_dt_fn=_hip_dt, full_sensors=False)
    ref = PropagationModel().StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    assert _engine.placement_cache_release() == 0
    pm = PropagationModel(keepPlacementCache=True)
    out = pm.StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    out2 = pm.StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    assert np.array_equal(out[2]['Pressure'], ref[2]['Pressure']) and np.array_equal(out2[2]['Pressure'], ref[2]['Pressure'])
    # an engine with arrays of another size: what the cache kept for the old size goes at bfd_create
    b, kb, infob = H.make_problem('C2', N=(80, 80, 64), steps=10, stable_dt_fn=_hip_dt, full_sensors=False)
    eng = _engine_for(b, kb, infob)
    held_other_size = _engine.placement_cache_release()
    eng.run(10)
    eng.close()
    assert held_other_size == 0
    assert 0 <= _engine.placement_cache_release() <= 1 << 30

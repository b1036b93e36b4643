"""The Z-slab split behind the drop-in call (bfd_group_*): ONE process, ONE call to
PropagationModel.StaggeredFDTD_3D_with_relaxation -- what BabelIntegrationBASE.py:2338-2365 does from the single child
process of Babel_SingleTx.py:258 -- runs on several slab engines whose halo planes move by peer copies inside the library.
On the 1-GPU box every slab sits on device 0 (an ordinal may repeat); the same call over distinct devices runs where they
exist. Every return value must equal the single-device call bit for bit, in the reference's own (C-order) array layout."""
import numpy as np
import pytest

from babelbrain_amd import PropagationModel, _engine, harness as H
from tests.util import oracle_dt

pytestmark = pytest.mark.gpu


def _same(out, ref):
    assert len(out) == len(ref)
    assert np.array_equal(out[-1]['IndexSensorMap'], ref[-1]['IndexSensorMap'])
    assert np.array_equal(out[0]['time'], ref[0]['time'])
    for q in range(len(ref) - 1):
        assert set(out[q].keys()) == set(ref[q].keys())
        for n in ref[q]:
            assert out[q][n].shape == ref[q][n].shape and out[q][n].dtype == ref[q][n].dtype, (q, n)
            assert np.array_equal(out[q][n], ref[q][n]), (q, n)


def _problem(config, N, steps, **kw):
    a, k, info = H.make_problem(config, N=N, steps=steps, stable_dt_fn=oracle_dt, **kw)
    k['SelMapsRMSPeakList'] = ['Pressure', 'Vz', 'Sigmazz', 'Sigmaxx']
    k['SelMapsSensorsList'] = ['Pressure', 'Sigmayy']
    k['SelRMSorPeak'] = 3
    return a, k, info


@pytest.mark.parametrize('config,nslabs,overlap', [('C2', 2, '1'), ('C2', 3, '0'), ('C3', 3, '1'), ('C1', 4, '1')])
def test_group_call_equals_single_device_call(config, nslabs, overlap, monkeypatch):
    """the wave crosses every interface (source plane k = 12, 32-plane slabs, 520 steps of 0.13-0.23 cells each)"""
    monkeypatch.setenv('BFD_GROUP_OVERLAP', overlap)
    a, k, info = _problem(config, (64, 56, 32 * nslabs), 520)
    ref = PropagationModel(device=0).StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    pm = PropagationModel(devices=[0] * nslabs)
    out = pm.StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    _same(out, ref)
    assert out[-1]['devices'] == [0] * nslabs and len(out[-1]['slabs']) == nslabs
    assert pm.last_timing['overlapped'] == (overlap == '1') and pm.last_timing['halo_bytes_per_step'] > 0
    # the wave did reach the last slab
    k0 = out[-1]['slabs'][-1][0]
    assert ref[2]['Pressure'][:, :, k0:].max() > 0


def test_group_through_the_reference_keyword_and_environment(monkeypatch):
    """DefaultGPUDeviceNumber=[...] in the call, or BABELFDTD_DEVICES for a caller that builds PModel without arguments
    (BASE:43); reflector mask, stress source, in-loop DFT instead of the series"""
    a, k, info = _problem('C2', (56, 64, 150), 300)
    refl = np.zeros(a[0].shape, np.uint32)
    refl[20:30, 20:28, 70:76] = 1
    k.update(ReflectorMask=refl, TypeSource=2, Ox=np.array([1.0]), Oy=np.array([1.0]), Oz=np.array([1.0]))
    ref = PropagationModel(device=0).StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, ReturnSensorDFT=True, ReturnSensorSeries=False, **k)
    out = PropagationModel().StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, DefaultGPUDeviceNumber=[0, 0], ReturnSensorDFT=True,
                                                               ReturnSensorSeries=False, **k)
    _same(out, ref)
    for n in ref[-1]['SensorDFT']:
        assert np.array_equal(out[-1]['SensorDFT'][n], ref[-1]['SensorDFT'][n]) and np.array_equal(out[-1]['SensorPeak'][n], ref[-1]['SensorPeak'][n])
    monkeypatch.setenv('BABELFDTD_DEVICES', '0,0,0')
    out = PropagationModel().StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, ReturnSensorDFT=True, ReturnSensorSeries=False, **k)
    _same(out, ref)
    assert len(out[-1]['slabs']) == 3


@pytest.mark.parametrize('seed', [3, 11])
def test_group_on_random_media(seed):
    """irregular solid islands, specks, reflector pockets, ragged grids (tests/test_random_media_gpu.py's generator)"""
    from tests.test_random_media_gpu import random_case
    a, k = random_case(seed)
    ref = PropagationModel(device=0).StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    out = PropagationModel(devices=[0, 0, 0]).StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    _same(out, ref)


def test_group_c_abi_with_x_fastest_views_and_reset():
    """the C ABI directly: Fortran-ordered (x-fastest) whole-domain views need no staging copy; reset + rerun reproduces"""
    a, k, info = _problem('C2', (64, 48, 140), 260)
    MaterialMap, ml, f, SourceMap, Pulse, h, T, SensorMap = a
    ref = PropagationModel(device=0).StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    from babelbrain_amd.PropagationModel import compact_sources
    g = _engine.Group([0, 0], *MaterialMap.shape, len(ml), h, k['DT'], f, info['nt'], sensorSub=k['SensorSubSampling'],
                      sensorStart=k['SensorStart'], selRMSorPeak=3, selMapsRMS=k['SelMapsRMSPeakList'], selMapsSensors=k['SelMapsSensorsList'])
    try:
        g.set_materials(ml, k.get('QCorrection', 1.0))
        g.set_material_map(np.asfortranarray(MaterialMap))
        lin, row, wx, wy, wz = compact_sources(SourceMap, k['Ox'], k['Oy'], k['Oz'])
        g.set_sources(lin.astype(np.int64), row, wx, wy, wz, Pulse)
        n = g.set_sensor_map(np.asfortranarray(SensorMap))
        assert n == ref[-1]['IndexSensorMap'].size
        for rep in range(2):
            g.run(info['nt'])
            g.sync()
            out = np.zeros(MaterialMap.shape, np.float32, order='F')
            g.get_map(_engine.KIND_RMS, 'Pressure', out)
            assert np.array_equal(out, ref[2]['Pressure'])
            assert np.array_equal(g.sensors()[g.selS.index('Pressure')], ref[0]['Pressure'])
            assert np.array_equal(g.sensor_index(), ref[-1]['IndexSensorMap'])
            g.reset()
        k0, nk, dev, view = g.slab(1)
        assert (k0, nk, dev) == (70, 70, 0) and view.tile_counts()['solid'] >= 0
    finally:
        g.close()


def test_group_halo_planes_through_the_peer_copy_call(monkeypatch):
    """BFD_GROUP_FORCE_PEER_COPY=1: the halo planes of slabs that share the device move through hipMemcpyPeerAsync, the call
    (and argument order) the path over distinct GPUs makes, in both step orders and with the per-slab host threads."""
    monkeypatch.setenv('BFD_GROUP_FORCE_PEER_COPY', '1')
    a, k, info = _problem('C2', (64, 56, 96), 520)
    ref = PropagationModel(device=0).StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    for overlap, threads in (('1', '1'), ('0', '1'), ('1', '0')):
        monkeypatch.setenv('BFD_GROUP_OVERLAP', overlap)
        monkeypatch.setenv('BFD_GROUP_THREADS', threads)
        out = PropagationModel(devices=[0, 0, 0]).StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
        _same(out, ref)
    assert ref[2]['Pressure'][:, :, 64:].max() > 0


def test_group_over_distinct_devices():
    """the same call over >= 2 GPUs (peer copies over xGMI); skips on the 1-GPU box"""
    devs = [d for d, _ in _engine.list_devices()]
    if len(devs) < 2:
        pytest.skip('needs two GPUs')
    a, k, info = _problem('C2', (64, 56, 64 * len(devs)), 250 + 480 * (len(devs) - 1))
    ref = PropagationModel(device=0).StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    out = PropagationModel(devices=devs).StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    _same(out, ref)
    assert ref[2]['Pressure'][:, :, out[-1]['slabs'][-1][0]:].max() > 0
    # every interface must have gone device to device: a missing peer path (staged through the host) fails nothing but the curve
    peer = out[-1]['timing']['peer']
    assert len(peer) == len(devs) - 1 and all(p['direct'] and all(p['can_access']) and all(p['enabled']) for p in peer), peer


def test_group_reports_how_halo_planes_travel():
    """bfd_group_peer_status through the drop-in call: slabs that share a device report a device copy on every interface."""
    a, k, info = _problem('C2', (64, 56, 128), 40)
    out = PropagationModel(devices=[0, 0, 0]).StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    peer = out[-1]['timing']['peer']
    assert [p['interface'] for p in peer] == [0, 1] and all(p['path'].startswith('same device') and p['direct'] for p in peer), peer
    assert all(p['devices'] == [0, 0] for p in peer)


def test_group_slabs_prepared_side_by_side(monkeypatch):
    """One slab per device prepares all slabs at once on their host threads (each device's only tenant runs the placement search a single-device
    call runs; eight searches then cost the time of one). On the 1-GPU box BFD_GROUP_PARALLEL_PREPARE=1 sends slabs that share the device
    through the same threads -- with the placement probe forced onto this small grid -- and the call must equal the single-device one."""
    monkeypatch.setenv('BFD_GROUP_PARALLEL_PREPARE', '1')
    monkeypatch.setenv('BFD_PLACEMENT_MIN_VOXELS', '0')
    a, k, info = _problem('C2', (64, 56, 128), 300)
    ref = PropagationModel(device=0).StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    out = PropagationModel(devices=[0, 0, 0, 0]).StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    _same(out, ref)
    assert len(out[-1]['placement']) == 4 and ref[2]['Pressure'].max() > 0

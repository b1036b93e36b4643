"""Parity of the HIP engine (through the C ABI / ctypes shim) against the CPU oracle.

Tolerance: BASELINE.json north_star -- pressure fields within 1e-5 relative L2 on identical
inputs. (Both sides follow the same canonical float32 arithmetic, so the observed error is 0 or
a few ulp; the assertion keeps the stated 1e-5.)
"""
import numpy as np
import pytest

from babelbrain_amd import harness as H
from oracle import oracle as O
from tests.util import ALL_MAPS, compare_runs, oracle_dt, rel_l2

pytestmark = pytest.mark.gpu
TOL = 1e-5


def hip_model(variant=0):
    from babelbrain_amd import PropagationModel
    return PropagationModel(kernelVariant=variant)


def run_both(args, kwargs, variant=0):
    out_h = hip_model(variant).StaggeredFDTD_3D_with_relaxation(*args, SILENT=True, **kwargs)
    out_r = O.StaggeredFDTD_3D_with_relaxation(*args, **kwargs)
    return out_h, out_r


@pytest.mark.parametrize('variant', [1, 2, 3])
def test_water_c1_small(variant):
    a, k, info = H.make_problem('C1', N=(48, 52, 64), steps=160, stable_dt_fn=oracle_dt)
    k['SelMapsRMSPeakList'] = ALL_MAPS
    oh, orf = run_both(a, k, variant)
    w = compare_runs(oh, orf, TOL)
    assert orf[2]['Pressure'].max() > 0
    print('worst rel L2', w)


@pytest.mark.parametrize('variant', [1, 2, 3])
def test_skull3_c2_small(variant):
    a, k, info = H.make_problem('C2', N=(64, 60, 72), steps=220, stable_dt_fn=oracle_dt)
    k['SelMapsRMSPeakList'] = ALL_MAPS
    k['SelMapsSensorsList'] = ['Pressure', 'Vz', 'Sigmaxy']
    k['SelRMSorPeak'] = 3
    oh, orf = run_both(a, k, variant)
    w = compare_runs(oh, orf, TOL, both=True)
    assert np.abs(orf[1]['Sigmaxy']).max() > 0, 'shear never excited: test is not covering the solid path'
    print('worst rel L2', w)


@pytest.mark.parametrize('variant', [1, 2, 3])
def test_ct_bins_qcorr_reflector(variant):
    a, k, info = H.make_problem('C3', N=(56, 56, 70), steps=200, stable_dt_fn=oracle_dt)
    mm = a[0]
    refl = np.zeros(mm.shape, np.uint32)
    refl[20:26, 30:34, 40:44] = 1            # an "air" pocket (BASE:2182-2190)
    k['ReflectorMask'] = refl
    k['SelMapsRMSPeakList'] = ['Pressure', 'Vx', 'Sigmazz', 'ALLV']
    oh, orf = run_both(a, k, variant)
    compare_runs(oh, orf, TOL)
    assert np.all(oh[1]['Pressure'][refl > 0] == 0)


@pytest.mark.parametrize('variant', [1, 2, 3])
def test_stress_point_source_backprop(variant):
    """Second solver call of the reference (BASE:2374-2398): point stress source, plane sensor,
    Ox/Oy/Oz left at their size-1 defaults."""
    a, k, info = H.make_problem('C2', N=(50, 45, 61), steps=150, stable_dt_fn=oracle_dt)
    mm, ml, f, smap, pulse, h, T, sensor = a
    N1, N2, N3 = mm.shape
    smap = np.zeros_like(smap)
    smap[N1 // 2, N2 // 2, N3 // 2] = 1
    pulse = H.punctual_source(f, k['DT'], T, ramp_length=1)
    _, back = H.sensor_maps(N1, N2, N3, info['zsrc'])
    for key in ('Ox', 'Oy', 'Oz'):
        k.pop(key)
    k['TypeSource'] = 2
    oh, orf = run_both((mm, ml, f, smap, pulse, h, T, back), k, variant)
    compare_runs(oh, orf, TOL)
    i, j, kk = H.decode_sensor_index(oh[-1]['IndexSensorMap'], N1, N2)
    assert np.all(kk == H.PML_THICKNESS)       # asserted by the caller too, BASE:2537
    assert np.abs(oh[0]['Pressure']).max() > 0


def test_hard_sources_and_no_sensors():
    a, k, info = H.make_problem('C1', N=(40, 40, 44), steps=60, stable_dt_fn=oracle_dt)
    a = list(a)
    a[7] = np.zeros_like(a[7])                 # empty SensorMap
    k['TypeSource'] = 1
    oh, orf = run_both(tuple(a), k)
    compare_runs(oh, orf, TOL)
    assert oh[0]['Pressure'].shape == (0, orf[0]['Pressure'].shape[1])


def test_no_sources_stays_zero():
    a, k, info = H.make_problem('C1', N=(40, 40, 44), steps=20, stable_dt_fn=oracle_dt)
    a = list(a)
    a[3] = np.zeros_like(a[3])
    out = hip_model().StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    assert out[2]['Pressure'].max() == 0 and np.all(out[0]['Pressure'] == 0)


def test_stable_dt_and_tables_match_oracle():
    from babelbrain_amd import _engine
    ml = H.ct_material_rows(500e3, 64)
    h = H.spatial_step(500e3, 6)
    q = np.ones(len(ml)); q[2:] = 3
    for corr in (True, False):
        dt_h = _engine.stable_dt(ml, 500e3, corr, h, 0.5, q)
        dt_o = O.stable_dt(ml, 500e3, corr, h, 0.5, q)
        assert dt_h == dt_o
        th, ch, cmh = _engine.material_tables(ml, 500e3, corr, h, dt_h, q)
        to, co, cmo = O.tables(ml, 500e3, h, dt_o, corr, q)
        assert np.array_equal(th, to) and np.array_equal(ch, co) and cmh == cmo


def test_calculate_matrices_tuple():
    pm = hip_model()
    ml = np.array([H.MATERIALS[500e3][n] for n in ('Water', 'Cortical', 'Brain')])
    dummy = np.zeros((10, 10, 3), np.uint32)
    out = pm.CalculateMatricesForPropagation(dummy, ml, 500e3, True, 3.675e-4, 0.5)
    assert len(out) == 10 and out[0] > 0          # unpacked as 10 values at BASE:1799
    out1 = pm.CalculateMatricesForPropagation(dummy * 0, ml[0, :].reshape((1, 5)), 500e3, True, 3.675e-4, 1.0)
    assert out1[0] > out[0]


def test_error_paths_raise():
    from babelbrain_amd._engine import EngineError
    a, k, info = H.make_problem('C1', N=(40, 40, 44), steps=10, stable_dt_fn=oracle_dt)
    pm = hip_model()
    bad = list(a); bad[0] = a[0] + 5                  # material id beyond MaterialList
    with pytest.raises(EngineError):
        pm.StaggeredFDTD_3D_with_relaxation(*bad, SILENT=True, **k)
    k2 = dict(k); k2['DT'] = k['DT'] * 20             # unstable time step
    with pytest.raises(EngineError):
        pm.StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k2)
    with pytest.raises(EngineError):
        pm.StaggeredFDTD_3D_with_relaxation(a[0][:20, :20, :20], *a[1:3], a[3][:20, :20, :20], *a[4:7], a[7][:20, :20, :20],
                                            SILENT=True, **{kk: v for kk, v in k.items() if kk not in ('Ox', 'Oy', 'Oz')})


def test_collapsed_fluid_slab_and_expansion():
    """All-fluid slab with no per-component stress output: the engine keeps only Szz/Rzz of the three
    identical normal stresses (DESIGN.md 'Tile classes'); outputs and the expanded fields must still
    match the oracle."""
    from babelbrain_amd import _engine
    from babelbrain_amd.PropagationModel import compact_sources
    a, k, info = H.make_problem('C3', N=(64, 48, 70), steps=150, stable_dt_fn=oracle_dt)
    k['SelMapsRMSPeakList'] = ['Pressure', 'Vx']
    k['SelRMSorPeak'] = 3
    oh, orf = run_both(a, k, 3)
    compare_runs(oh, orf, TOL, both=True)
    # engine level: the collapsed flag is on, and get_field expands Sxx/Syy from Szz
    mm, ml, f, smap, pulse, h, T, sensor = a
    N1, N2, N3 = mm.shape
    eng = _engine.Engine(N1, N2, N3, len(ml), h, k['DT'], f, info['nt'], sensorSub=k['SensorSubSampling'],
                         sensorStart=k['SensorStart'], selMapsRMS=['Pressure'], selMapsSensors=['Pressure'], kernelVariant=3)
    eng.set_materials(ml, k['QCorrection'])
    eng.set_material_map(mm, 0, 0)
    eng.set_sources(*compact_sources(smap, k['Ox'], k['Oy'], k['Oz']), pulse)
    eng.set_sensor_map(sensor)
    assert eng.tile_counts()['solid'] == 0
    eng.run(info['nt'])
    szz = eng.get_field('Szz')
    assert np.abs(szz).max() > 0
    assert np.array_equal(eng.get_field('Sxx'), szz) and np.array_equal(eng.get_field('Syy'), szz)
    assert np.array_equal(eng.get_field('Rxx'), eng.get_field('Rzz'))
    k2 = dict(k); k2['SelMapsRMSPeakList'] = ['Sigmaxx', 'Sigmazz']; k2['SelRMSorPeak'] = 1
    ref = O.StaggeredFDTD_3D_with_relaxation(*a, **k2)
    assert rel_l2(szz, ref[1]['Sigmazz']) <= TOL and rel_l2(eng.get_field('Sxx'), ref[1]['Sigmaxx']) <= TOL
    eng.close()


def test_lean_fluid_tiles_beside_solid_ones():
    """Slab with solid tiles: every fluid CELL keeps a single copy of its identical normal stresses (class byte,
    bfd_dev::cls), in fluid tiles and inside solid runs alike, whatever outputs are selected; Pressure outputs
    (sensors, RMS, peak, last map), per-component stress maps and the fields expanded on demand must match the oracle
    everywhere."""
    from babelbrain_amd import _engine
    from babelbrain_amd.PropagationModel import compact_sources
    a, k, info = H.make_problem('C2', N=(136, 60, 72), steps=200, stable_dt_fn=oracle_dt)
    k['SelMapsRMSPeakList'] = ['Pressure', 'Vz']
    k['SelMapsSensorsList'] = ['Pressure', 'Vx']
    k['SelRMSorPeak'] = 3
    oh, orf = run_both(a, k, 3)
    compare_runs(oh, orf, TOL, both=True)
    mm, ml, f, smap, pulse, h, T, sensor = a
    N1, N2, N3 = mm.shape
    eng = _engine.Engine(N1, N2, N3, len(ml), h, k['DT'], f, info['nt'], sensorSub=k['SensorSubSampling'],
                         sensorStart=k['SensorStart'], selMapsRMS=['Pressure'], selMapsSensors=['Pressure'], kernelVariant=3)
    eng.set_materials(ml, k['QCorrection'])
    eng.set_material_map(mm, 0, 0)
    eng.set_sources(*compact_sources(smap, k['Ox'], k['Oy'], k['Oz']), pulse)
    eng.set_sensor_map(sensor)
    tc = eng.tile_counts()
    assert tc['solid'] > 0 and 0 < tc['lean_fluid'] == tc['lossless_fluid'] + tc['lossy_fluid'], tc
    eng.run(info['nt'])
    k2 = dict(k); k2['SelMapsRMSPeakList'] = ['Sigmaxx', 'Sigmayy', 'Sigmazz', 'Sigmaxy']; k2['SelRMSorPeak'] = 1
    ref = O.StaggeredFDTD_3D_with_relaxation(*a, **k2)
    for name, key in (('Sxx', 'Sigmaxx'), ('Syy', 'Sigmayy'), ('Szz', 'Sigmazz'), ('Sxy', 'Sigmaxy')):
        got = eng.get_field(name)
        assert np.abs(ref[1][key]).max() > 0
        assert rel_l2(got, ref[1][key]) <= TOL, name
    eng.close()
    # per-component outputs read the single copy where the cell is fluid
    k3 = dict(k); k3['SelMapsRMSPeakList'] = ['Sigmaxx', 'Sigmayy', 'Sigmazz', 'Sigmaxy', 'Pressure']; k3['SelMapsSensorsList'] = ['Sigmaxx', 'Sigmayy']
    oh, orf = run_both(a, k3, 3)
    compare_runs(oh, orf, TOL, both=True)


@pytest.mark.parametrize('variant', [1, 2, 3])
@pytest.mark.parametrize('N', [(30, 30, 30), (31, 66, 33), (130, 30, 40)])
def test_smallest_and_ragged_grids(variant, N):
    """Edge sizes: the smallest grid the engine accepts (2*(NDelta+1)+4 = 30 per axis; smaller ones are refused, see
    test_error_paths_raise), tile-ragged in x (130 -> a third tile with 2 live columns) and in y; a point source in the only interior cell region, every cell a sensor, solid
    block touching the layer."""
    N1, N2, N3 = N
    rng = np.random.default_rng(N1 * 1000 + N2)
    h = 1102.515 / 500e3 / 6
    M = H.MATERIALS[500e3]
    ml = np.array([M['Water'], M['Cortical'], M['Brain']], float)
    mm = np.zeros(N, np.uint32)
    mm[N1 // 2:, :, N3 // 2:] = 1
    mm[:N1 // 3, N2 // 2:, :] = 2
    dt = oracle_dt(ml, 500e3, h, 0.9)
    nt = 40
    smap = np.zeros(N, np.uint32)
    smap[13, 13, 13] = 1
    smap[N1 - 14, N2 - 14, 13] = 2
    pulse = np.sin(2 * np.pi * 500e3 * dt * np.arange(nt + 1))[None, :] * np.array([[1.0], [0.5]])
    sensor = np.ones(N, np.uint32)
    Oz = rng.uniform(0.5, 1.0, N) / 1.5e6
    kw = dict(Ox=np.zeros(N), Oy=rng.uniform(0, 1, N) / 1.5e6, Oz=Oz, NDelta=12, DT=dt, SensorSubSampling=3, SensorStart=2,
              SelMapsRMSPeakList=['Pressure', 'Vx', 'Sigmaxy'], SelMapsSensorsList=['Pressure'], SelRMSorPeak=3, TypeSource=0,
              QfactorCorrection=True, QCorrection=1.0)
    oh, orf = run_both((mm, ml, 500e3, smap, pulse, h, nt * dt, sensor), kw, variant)
    compare_runs(oh, orf, TOL, both=True)
    assert np.abs(orf[0]['Pressure']).max() > 0
    assert oh[0]['Pressure'].shape[0] == N1 * N2 * N3


def test_graph_replay_equals_direct_launches(monkeypatch):
    """With BFD_USE_GRAPH=1 bfd_run replays the plain steps (before the accumulation / sensor window) from a hipGraph
    of 8 steps with a device-side step counter (off by default: measured slower than direct launches). Same results
    bit for bit, also when the run is cut into pieces and when the inputs are replaced in between."""
    from babelbrain_amd import _engine
    from babelbrain_amd.PropagationModel import compact_sources
    a, k, info = H.make_problem('C2', N=(64, 60, 72), steps=230, stable_dt_fn=oracle_dt)
    mm, ml, f, smap, pulse, h, T, sensor = a
    N1, N2, N3 = mm.shape
    assert k['SensorStart'] * k['SensorSubSampling'] > 64          # plenty of plain steps at the start

    def run(pieces, src_scale=1.0):
        eng = _engine.Engine(N1, N2, N3, len(ml), h, k['DT'], f, info['nt'], sensorSub=k['SensorSubSampling'],
                             sensorStart=k['SensorStart'], selMapsRMS=['Pressure'], selMapsSensors=['Pressure', 'Vz'], kernelVariant=3)
        eng.set_materials(ml, k['QCorrection'])
        eng.set_material_map(mm, 0, 0)
        eng.set_sources(*compact_sources(smap, k['Ox'], k['Oy'], k['Oz']), pulse)
        eng.set_sensor_map(sensor)
        done = 0
        for i, n in enumerate(pieces):
            if i == 1 and src_scale != 1.0:
                eng.set_sources(*compact_sources(smap, k['Ox'], k['Oy'], k['Oz']), pulse * src_scale)
            eng.run(n)
            done += n
        assert done == info['nt'] and eng.step == info['nt']
        out = (eng.sensors().copy(), eng.get_map(_engine.KIND_RMS, 'Pressure'), eng.get_field('Vz'), eng.get_field('Sxy'))
        eng.close()
        return out

    ref = None
    for use_graph in (None, '1'):
        if use_graph:
            monkeypatch.setenv('BFD_USE_GRAPH', use_graph)
        else:
            monkeypatch.delenv('BFD_USE_GRAPH', raising=False)
        whole = run([info['nt']])
        parts = run([19, 3, 40, 1, info['nt'] - 63])
        scaled = run([30, info['nt'] - 30], src_scale=0.5)
        for x, y in zip(whole, parts):
            assert np.array_equal(x, y)
        if ref is None:
            ref = (whole, scaled)
            assert np.abs(whole[1]).max() > 0 and not np.array_equal(whole[1], scaled[1])
        else:
            for x, y in zip(ref[0], whole):
                assert np.array_equal(x, y)
            for x, y in zip(ref[1], scaled):
                assert np.array_equal(x, y)


@pytest.mark.parametrize('config', ['C1', 'C1-lossy', 'C2', 'C3', 'C3-lossless', 'C3-peak', 'C1-rms'])
def test_fused_fluid_step_variant4(config):
    """kernelVariant 4: eligible fluid runs (64 x 24 cells, bfd_kernels_fused.hip) advance both half-steps in one pass over
    two copies of V, Szz, Rzz; everything else runs the variant-3 kernels out of place. Outputs and raw fields must equal
    the oracle's like every other variant. The grid holds two columns of fused tiles in x, two groups of three tile rows and
    a left-over row in y, a short and a full z-run; the media cover the four flavours of the fused body (one material / ids
    per cell, with / without memory variables)."""
    from babelbrain_amd import _engine
    from babelbrain_amd.PropagationModel import compact_sources
    N = (214, 90, 80)
    a, k, info = H.make_problem(config.split('-')[0], N=N, steps=170, stable_dt_fn=oracle_dt)
    if config == 'C1-lossy':                                   # one attenuating fluid everywhere: the LOSSY flavour of the fused body
        a = list(a); a[1] = np.array([[1041.0, 1562.0, 0.0, 30.0, 0.0]]); a = tuple(a)
        assert k['DT'] <= oracle_dt(a[1], a[2], a[5], 0.95)
    if config == 'C3-lossless':                                # many fluids, none attenuating: ids per cell, no memory variables
        a = list(a); ml = np.array(a[1], np.float64); ml[:, 3:] = 0.0; a[1] = ml; a = tuple(a)
    k['SelMapsRMSPeakList'] = ['Pressure', 'Vx', 'Vz']
    k['SelMapsSensorsList'] = ['Pressure', 'Vy']
    k['SelRMSorPeak'] = {'C3-peak': 2, 'C1-rms': 1}.get(config, 3)      # the fused body has a flavour per accumulation mode (sums / peaks / both)
    oh, orf = run_both(a, k, 4)
    compare_runs(oh, orf, TOL, both=(k['SelRMSorPeak'] == 3))
    assert orf[2]['Pressure'].max() > 0
    mm, ml, f, smap, pulse, h, T, sensor = a
    N1, N2, N3 = mm.shape
    eng = _engine.Engine(N1, N2, N3, len(ml), h, k['DT'], f, info['nt'], sensorSub=k['SensorSubSampling'],
                         sensorStart=k['SensorStart'], selMapsRMS=['Pressure'], selMapsSensors=['Pressure'], kernelVariant=4)
    eng.set_materials(ml, k['QCorrection'])
    eng.set_material_map(mm, 0, 0)
    eng.set_sources(*compact_sources(smap, k['Ox'], k['Oy'], k['Oz']), pulse)
    eng.set_sensor_map(sensor)
    tc = eng.tile_counts()
    print(config, tc)
    assert tc['fused_fluid'] > 0 or config == 'C2', tc          # C2's shell at this size leaves no eligible run; its ping-pong path is still covered
    if config == 'C1-lossy':
        assert tc['lossy_fluid'] > 0 and tc['lossless_fluid'] == 0
    eng.run(info['nt'] - 1)
    eng.half_step_stress(); eng.half_step_velocity()         # odd and even numbers of swaps both leave the right copy current
    k2 = dict(k); k2['SelMapsRMSPeakList'] = ['Sigmaxx', 'Sigmazz', 'Vx', 'Vy', 'Vz']; k2['SelRMSorPeak'] = 1
    ref = O.StaggeredFDTD_3D_with_relaxation(*a, **k2)
    for name, key in (('Szz', 'Sigmazz'), ('Sxx', 'Sigmaxx'), ('Vx', 'Vx'), ('Vy', 'Vy'), ('Vz', 'Vz')):
        assert rel_l2(eng.get_field(name), ref[1][key]) <= TOL, name
    with pytest.raises(_engine.EngineError):
        eng.half_step_stress(1)                                # split half-steps belong to Z-slabs (which stay in place)
    eng.close()


@pytest.mark.parametrize('variant', [1, 3])
@pytest.mark.parametrize('nd,rl', [(8, 1e-4), (16, 1e-6)])
def test_other_absorbing_layer_settings(variant, nd, rl):
    """NDelta and ReflectionLimit are arguments of the call (BASE:2350, 2352); the tile classes and the compact
    CPML storage follow the layer width."""
    a, k, info = H.make_problem('C2', N=(72, 64, 80), steps=160, stable_dt_fn=oracle_dt)
    k['NDelta'] = nd
    k['ReflectionLimit'] = rl
    k['SelMapsRMSPeakList'] = ['Pressure', 'Vx', 'Sigmaxz']
    oh, orf = run_both(a, k, variant)
    compare_runs(oh, orf, TOL)
    assert orf[2]['Pressure'].max() > 0


@pytest.mark.parametrize('tile,type_source', [(16, 0), (7, 2), (1000, 0)])
def test_streamed_source_table_equals_resident(monkeypatch, tile, type_source):
    """Large PulseSource tables stay on the host and reach the device in double-buffered time tiles (float64 -> float32
    on the way); forced here with BFD_SOURCE_TILE. Every output equals the run with the resident table bit for bit --
    also after a reset (the tiles start over) and for stress sources (injected after the stress half-step)."""
    from babelbrain_amd import PropagationModel, _engine
    from babelbrain_amd.PropagationModel import compact_sources
    a, k, info = H.make_problem('C2', N=(48, 40, 56), steps=150, stable_dt_fn=oracle_dt)
    k['TypeSource'] = type_source
    k['SelMapsRMSPeakList'] = ['Pressure', 'Vz']
    ref = PropagationModel().StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    monkeypatch.setenv('BFD_SOURCE_TILE', str(tile))
    got = PropagationModel().StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    assert got[-1]['device_bytes'] != ref[-1]['device_bytes']
    for name in ('Pressure', 'Vz'):
        assert np.array_equal(got[2][name], ref[2][name]) and np.array_equal(got[1][name], ref[1][name])
    assert np.array_equal(got[0]['Pressure'], ref[0]['Pressure'])
    # engine level: run, reset, run again
    mm, ml, f, smap, pulse, h, T, sensor = a
    eng = _engine.Engine(*mm.shape, len(ml), h, k['DT'], f, info['nt'], typeSource=type_source, sensorSub=k['SensorSubSampling'],
                         sensorStart=k['SensorStart'], selMapsRMS=['Pressure'], selMapsSensors=['Pressure'])
    eng.set_materials(ml, k['QCorrection'])
    eng.set_material_map(mm, 0, 0)
    pulse64 = np.ascontiguousarray(pulse, np.float64)
    eng.set_sources(*compact_sources(smap, k['Ox'], k['Oy'], k['Oz']), pulse64)
    eng.set_sensor_map(sensor)
    eng.run(60)
    eng.reset()
    eng.run(info['nt'])
    assert np.array_equal(eng.get_map(_engine.KIND_RMS, 'Pressure'), ref[2]['Pressure'])
    eng.close()


def test_placement_choice_does_not_change_results(monkeypatch):
    """bfd_prepare probes pairs of arrays on the zero state (a += b, b += a along the run lists), exchanges buffers between the
    state arrays and allocates fresh ones where a memory region is short (grids of 32 M voxels and more by
    default, forced here with a short search). The run that follows must be the run without it: the probes leave the state
    untouched (also the peak map), exchanged and fresh buffers are zero like the ones they replace."""
    monkeypatch.setenv('BFD_PLACEMENT_MIN_VOXELS', '0')
    monkeypatch.setenv('BFD_PLACEMENT_SEARCH_MB', '300')
    a, k, info = H.make_problem('C2', N=(192, 160, 160), steps=60, stable_dt_fn=oracle_dt)
    k['SelMapsRMSPeakList'] = ['Pressure', 'Vz', 'Sigmaxy']
    k['SelRMSorPeak'] = 3
    monkeypatch.setenv('BFD_PLACEMENT', '0')
    ref = hip_model().StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    monkeypatch.setenv('BFD_PLACEMENT', '1')
    out = hip_model().StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    for idx in range(4):
        for name in ref[idx]:
            assert np.array_equal(ref[idx][name], out[idx][name]), (idx, name)
    assert np.abs(ref[1]['Sigmaxy']).max() > 0
    # the dense variant moves all 15 arrays and the class bytes
    out2 = hip_model(2).StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    for idx in range(4):
        for name in ref[idx]:
            assert np.array_equal(ref[idx][name], out2[idx][name]), (idx, name)
    # the fused fluid step keeps second copies of Vx Vy Vz Szz Rzz: they are placed (and exchanged) with the others
    out4 = hip_model(4).StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    for idx in range(4):
        for name in ref[idx]:
            assert np.array_equal(ref[idx][name], out4[idx][name]), (idx, name)


def test_cost_balanced_run_maps_do_not_change_results(monkeypatch):
    """BFD_XCD_BALANCE=1 (experiment): the run lists are cut among the 8 XCDs by estimated cost instead of by count
    and the launches carry surplus blocks that return at once. Which block takes which run must not matter: same bits, on a
    medium with solid runs in and outside the absorbing layer, also as the boundary / interior parts of a Z-slab."""
    from babelbrain_amd import slab
    from tests.test_slab_gpu import _exchange
    a, k, info = H.make_problem('C2', N=(192, 96, 96), steps=90, stable_dt_fn=oracle_dt)
    k['SelMapsRMSPeakList'] = ['Pressure', 'Vz', 'Sigmaxy']
    k['SelRMSorPeak'] = 3
    monkeypatch.setenv('BFD_XCD_BALANCE', '0')
    ref = hip_model().StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    monkeypatch.setenv('BFD_XCD_BALANCE', '1')
    out = hip_model().StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    for idx in range(4):
        for name in ref[idx]:
            assert np.array_equal(ref[idx][name], out[idx][name]), (idx, name)
    assert np.abs(ref[1]['Sigmaxy']).max() > 0
    # two slabs, split half-steps (parts 1 and 2 have their own maps)
    import torch
    from babelbrain_amd._engine import HALO_STRESS, HALO_VELOCITY
    slabs, infos = zip(*[slab.create_hip_slab(a, k, r, 2, 0, kernelVariant=0) for r in range(2)])
    for _ in range(info['nt']):
        for s in slabs: s.half_step_stress(1)
        _exchange(slabs, HALO_STRESS)
        for s in slabs: s.half_step_stress(2)
        for s in slabs: s.half_step_velocity(1)
        _exchange(slabs, HALO_VELOCITY)
        for s in slabs: s.half_step_velocity(2)
    torch.cuda.synchronize()
    merged = slab.merge_slab_outputs([slab.collect_slab_outputs(s.eng, k, i) for s, i in zip(slabs, infos)])
    for n in ref[2]:
        assert np.array_equal(merged['RMS'][n], ref[2][n]), n
        assert np.array_equal(merged['Peak'][n], ref[3][n]), n
    for s in slabs:
        s.eng.close()


def test_large_result_blocks_through_threads_equal_the_plain_copy(monkeypatch):
    """Sensor series and maps of 256 MB and more leave the device through several host threads with pinned pieces of their own
    (copy_out_large, bfd_api.hip). Here the same code on a small grid (thresholds lowered): every output equal to the one-copy path,
    for an even and an odd number of threads, single engine and two slabs (row pitch)."""
    a, k, info = H.make_problem('C2', N=(192, 160, 160), steps=80, stable_dt_fn=oracle_dt)
    k['SelMapsRMSPeakList'] = ['Pressure', 'Vz', 'Sigmaxy']
    k['SelMapsSensorsList'] = ['Pressure', 'Vx', 'Sigmaxz']
    k['SelRMSorPeak'] = 3
    monkeypatch.setenv('BFD_D2H_THREADS', '0')
    ref = hip_model().StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    assert ref[0]['Vx'].nbytes > (8 << 20) and np.abs(ref[0]['Vx']).max() > 0
    monkeypatch.setenv('BFD_D2H_MIN_MB', '1')
    monkeypatch.setenv('BFD_D2H_PIECE_KB', '256')
    from babelbrain_amd import PropagationModel
    for threads, devices in (('4', None), ('3', None), ('4', [0, 0])):
        monkeypatch.setenv('BFD_D2H_THREADS', threads)
        pm = hip_model() if devices is None else PropagationModel(devices=devices)
        out = pm.StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
        for idx in range(4):
            for name in ref[idx]:
                assert np.array_equal(ref[idx][name], out[idx][name]), (threads, devices, idx, name)

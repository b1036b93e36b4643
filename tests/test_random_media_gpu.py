"""Parity on randomised media: the structured test media (shell, CT bins, plates) leave most per-cell class patterns
unvisited -- single solid voxels in water, one-voxel fluid holes in bone, solid specks inside the absorbing layer, edges
whose four cells are solid only along one diagonal. Every case here draws its grid size, absorbing-layer width, material
map (smoothed-noise islands plus speckle), reflector pocket, source kind and placement, sensor set and map selection from a
seeded generator and holds the HIP engine to the oracle on every output (1e-5 rel-L2 as everywhere; observed 0).
The default variant is compared for all seeds; the dense and the simple variants for the first ones."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.util import ALL_MAPS, compare_runs, oracle_dt

pytestmark = pytest.mark.gpu

#            rho     cL     cS    aL    aS
MATERIALS = [[1000., 1500., 0., 0., 0.],          # water
             [1896.5, 2476., 1542., 81., 164.],    # cortical bone (shear, lossy)
             [1041., 1562., 0., 3.45, 0.],         # brain (fluid, lossy)
             [1738., 2205., 1313., 81., 164.],     # trabecular bone
             [2200., 3000., 1600., 0., 0.],        # a lossless solid
             [1116., 1537., 0., 2.3, 0.]]          # skin


def smooth(a, n):
    for _ in range(n):
        a = (a + np.roll(a, 1, 0) + np.roll(a, -1, 0) + np.roll(a, 1, 1) + np.roll(a, -1, 1) + np.roll(a, 1, 2) + np.roll(a, -1, 2)) / 7.0
    return a


def random_case(seed):
    rng = np.random.default_rng(1000 + seed)
    nd = int(rng.choice([5, 8, 12]))
    N = tuple(int(v) for v in rng.integers(2 * (nd + 1) + 6, 84, 3))     # the engine wants 2 (NDelta + 1) + 4 cells per axis at least
    if seed % 4 == 0:
        N = (int(rng.integers(65, 140)), N[1], N[2])           # more than one tile in x, ragged
    freq = 500e3
    ml = np.array(MATERIALS, np.float64)
    h = 1102.515 / freq / 6
    # islands: two smoothed noise fields -> solids where high, lossy fluid where the second is high; then speckle
    f1 = smooth(rng.standard_normal(N), int(rng.integers(1, 4)))
    f2 = smooth(rng.standard_normal(N), 2)
    mm = np.zeros(N, np.uint32)
    mm[f2 > np.quantile(f2, 0.7)] = 2
    mm[f2 > np.quantile(f2, 0.93)] = 5
    q1 = np.quantile(f1, [0.6, 0.8, 0.93])
    mm[f1 > q1[0]] = 1
    mm[f1 > q1[1]] = 3
    mm[f1 > q1[2]] = 4
    speck = rng.random(N)
    mm[speck < 0.004] = 1                                       # single solid voxels anywhere (also inside the absorbing layer)
    mm[(speck > 0.996) & (mm == 1)] = 0                         # one-voxel fluid holes in bone
    if seed % 3 == 1:
        mm[:, :, : N[2] // 3] = 0                               # a fluid-only stretch: collapsed / lean tiles beside solid ones
    refl = None
    if seed % 2 == 0:
        refl = np.zeros(N, np.uint32)
        c = [int(rng.integers(nd + 1, max(n - nd - 5, nd + 2))) for n in N]
        refl[c[0]:c[0] + 3, c[1]:c[1] + 4, c[2]:c[2] + 2] = 1
    type_source = 2 if seed % 5 == 3 else 0
    src = np.zeros(N, np.uint32)
    if seed % 2 == 0:                                           # a source plane just inside the layer
        zs = nd
        ii, jj = np.meshgrid(np.arange(nd, N[0] - nd), np.arange(nd, N[1] - nd), indexing='ij')
        keep = rng.random(ii.shape) < 0.5
        src[ii[keep], jj[keep], zs] = np.arange(1, int(keep.sum()) + 1)
    else:                                                       # scattered voxels, some sharing a row of the pulse table
        npts = int(rng.integers(3, 40))
        for s in range(npts):
            p = [int(rng.integers(nd, n - nd)) for n in N]
            src[p[0], p[1], p[2]] = 1 + s % 7
    nsrc = int(src.max())
    dt = oracle_dt(ml, freq, h, 0.99)
    ppp = int(np.ceil(1.0 / (freq * dt)))
    dt = 1.0 / (freq * ppp)
    nt = int(rng.integers(90, 180))
    t = np.arange(nt + 1) * dt
    amp = 1.0 + rng.random(nsrc)
    ph = 2 * np.pi * rng.random(nsrc)
    pulse = amp[:, None] * np.sin(2 * np.pi * freq * t[None, :] + ph[:, None])
    ramp = min(len(t), 2 * ppp)
    pulse[:, :ramp] *= (0.5 * (1 - np.cos(np.pi * np.arange(ramp) / ramp)))[None, :]
    sens = (rng.random(N) < 0.05).astype(np.uint32)
    sens[:nd] = 0; sens[-nd:] = 0; sens[:, :nd] = 0; sens[:, -nd:] = 0; sens[:, :, :nd] = 0; sens[:, :, -nd:] = 0
    full = np.ones(N, np.float64)
    weights = dict(Ox=full * rng.random(), Oy=full * rng.random(), Oz=full) if seed % 3 else dict(Ox=np.array([0.3]), Oy=np.array([0.0]), Oz=np.array([1.0]))
    sub = int(rng.choice([1, 2, 5]))
    k = dict(NDelta=nd, DT=dt, ReflectionLimit=1e-5, USE_SINGLE=True, SelRMSorPeak=int(rng.choice([1, 2, 3])),
             SelMapsRMSPeakList=ALL_MAPS if seed % 2 else ['Pressure', 'Vz', 'Sigmaxy'],
             SelMapsSensorsList=['Pressure', 'Vx', 'Sigmaxz'] if seed % 2 else ['Pressure'],
             SensorSubSampling=sub, SensorStart=int(rng.integers(0, nt // sub // 2)), TypeSource=type_source,
             QfactorCorrection=True, QCorrection=[1.0, 3.0, 1.0, 2.0, 1.0, 1.0] if seed % 2 else 1.0, ReflectorMask=refl, **weights)
    a = (mm, ml, freq, src, pulse, h, nt * dt, sens)
    return a, k


@pytest.mark.parametrize('seed', range(14))
def test_random_media_default_variant(seed):
    from babelbrain_amd import PropagationModel
    a, k = random_case(seed)
    oh = PropagationModel().StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    orf = O.StaggeredFDTD_3D_with_relaxation(*a, **k)
    w = compare_runs(oh, orf, 1e-5, both=(k['SelRMSorPeak'] == 3))
    solid = np.isin(a[0], [1, 3, 4])
    assert solid.any() and (~solid).any() and np.abs(orf[1]['Pressure']).max() > 0
    assert all(np.isfinite(v).all() for v in orf[1].values())
    print('seed %d grid %s layer %d: worst rel L2 %.2e' % (seed, a[0].shape, k['NDelta'], w))


@pytest.mark.parametrize('variant', [1, 2])
@pytest.mark.parametrize('seed', [0, 1, 3])
def test_random_media_other_variants(seed, variant):
    from babelbrain_amd import PropagationModel
    a, k = random_case(seed)
    oh = PropagationModel(kernelVariant=variant).StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    orf = O.StaggeredFDTD_3D_with_relaxation(*a, **k)
    compare_runs(oh, orf, 1e-5, both=(k['SelRMSorPeak'] == 3))


@pytest.mark.parametrize('seed,world,split', [(0, 2, True), (1, 2, False), (4, 3, True), (5, 3, True), (8, 2, True), (9, 2, True), (12, 3, False)])
def test_random_media_in_slabs(seed, world, split):
    """The same media cut into Z-slabs on one GPU (halo planes moved by device copies, tests/test_slab_gpu.py): slab
    interfaces land anywhere in the islands, sources and reflector pockets; both step orders; equal to the single domain."""
    import torch
    from babelbrain_amd import PropagationModel, slab
    from babelbrain_amd._engine import HALO_STRESS, HALO_VELOCITY
    from tests.test_slab_gpu import _exchange
    a, k = random_case(seed)
    ref = PropagationModel().StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    nt = ref[-1]['nt']
    slabs, infos = zip(*[slab.create_hip_slab(a, k, r, world, 0, kernelVariant=0) for r in range(world)])
    for _ in range(nt):
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
    torch.cuda.synchronize()
    merged = slab.merge_slab_outputs([slab.collect_slab_outputs(s.eng, k, i) for s, i in zip(slabs, infos)])
    assert np.array_equal(merged['IndexSensorMap'], ref[-1]['IndexSensorMap'])
    for n in k['SelMapsSensorsList']:
        assert np.array_equal(merged['Sensor'][n], ref[0][n]), ('sensor', n)
    names = ref[1].keys()
    for n in names:
        assert np.array_equal(merged['LastMap'][n], ref[1][n]), ('last', n)
        if k['SelRMSorPeak'] & 1:
            assert np.array_equal(merged['RMS'][n], ref[2][n]), ('rms', n)
        if k['SelRMSorPeak'] & 2:
            assert np.array_equal(merged['Peak'][n], ref[-2][n]), ('peak', n)
    for s in slabs:
        s.eng.close()


@pytest.mark.parametrize('seed', [2, 3, 5, 8, 13])
def test_random_media_with_the_placement_choice_forced(seed, monkeypatch):
    """bfd_prepare runs pair probes on the zero state, exchanges the buffers of state arrays and draws fresh ones to spread the
    arrays over memory regions (grids of 32 M voxels and more by default). Forced here on the small random cases, with a short
    search for fresh buffers -- stress-type sources (seeds 3, 8, 13), peak maps, reflector pockets, all selected maps -- the run
    that follows must still equal the oracle bit for bit."""
    from babelbrain_amd import PropagationModel
    monkeypatch.setenv('BFD_PLACEMENT_MIN_VOXELS', '0')
    monkeypatch.setenv('BFD_PLACEMENT_SEARCH_MB', '40')
    a, k = random_case(seed)
    oh = PropagationModel().StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    orf = O.StaggeredFDTD_3D_with_relaxation(*a, **k)
    assert compare_runs(oh, orf, 0.0, both=(k['SelRMSorPeak'] == 3)) == 0.0


@pytest.mark.parametrize('seed', [1, 4, 6])
def test_compact_solid_state_equals_full_volume_arrays(seed, monkeypatch):
    """Round 5: with the compact solid state (default) Sxx, Syy, the shear stresses and Rxx, Ryy of the solid cells live in sparse-list
    order, the fluid stress kernel takes Szz / Rzz of the solid runs and the sparse kernel everything else a solid cell has. All 15
    state arrays, read back through bfd_get_field (which fills the full-volume arrays from the compact ones), must equal those of a
    run with BFD_COMPACT_SOLID=0 (full-volume arrays, stress_solid + stress_shear_sparse) -- also after a reset and a second run."""
    from babelbrain_amd import _engine
    from babelbrain_amd.PropagationModel import compact_sources
    a, k = random_case(seed)
    mm, ml, f, smap, pulse, h, T, sensor = a
    nt = int(round(T / k['DT']))

    def fields(env, hosted='1', again=False):
        monkeypatch.setenv('BFD_COMPACT_SOLID', env)
        monkeypatch.setenv('BFD_COMPACT_HOSTED', hosted)
        eng = _engine.Engine(*mm.shape, len(ml), h, k['DT'], f, nt, NDelta=k['NDelta'], typeSource=k['TypeSource'], sensorSub=k['SensorSubSampling'],
                             sensorStart=k['SensorStart'], selMapsRMS=['Pressure'], selMapsSensors=['Pressure', 'Sigmaxx', 'Sigmaxz'], selRMSorPeak=1)
        eng.set_materials(ml, k['QCorrection'])
        eng.set_material_map(mm, 0, 0)
        if k['ReflectorMask'] is not None:
            eng.set_reflector(k['ReflectorMask'])
        eng.set_sources(*compact_sources(smap, k['Ox'], k['Oy'], k['Oz']), pulse)
        eng.set_sensor_map(sens_map)
        out = []
        for rep in range(2):
            eng.run(nt // 2)
            out.append({n: eng.get_field(n).copy() for n in _engine.FIELD_NAMES})
            out[-1]['sensors'] = eng.sensors().copy()
            if again:                         # inputs set again in the middle of a run: the lists are rebuilt, the compact values must move into the new list
                eng.set_sources(*compact_sources(smap, k['Ox'], k['Oy'], k['Oz']), pulse)
                # between a setter and the next step the old list is still the one the compact arrays are ordered by: a field read in that
                # window must be the same field (round 5 handed out the hosting buffers' raw content here)
                for n in _engine.FIELD_NAMES:
                    assert np.array_equal(eng.get_field(n), out[-1][n]), ('get_field between a setter and the next step', n)
            eng.run(nt - nt // 2)             # get_field in the middle of a run must not disturb it
            out.append({n: eng.get_field(n).copy() for n in _engine.FIELD_NAMES})
            eng.reset()
        eng.close()
        return out

    sens_map = sensor
    full = fields('0')
    assert any(np.abs(full[1][n]).max() > 0 for n in ('Sxy', 'Sxz', 'Syz', 'Rxx', 'Rxy'))
    # the compact arrays inside the full-volume buffers of their fields (default), in a block of their own, and with the lists rebuilt mid-run
    for comp in (fields('1'), fields('1', hosted='0'), fields('1', again=True), fields('1', hosted='0', again=True)):
        for q, (x, y) in enumerate(zip(full, comp)):
            for n in x:
                assert np.array_equal(x[n], y[n]), (q, n)


@pytest.mark.parametrize('seed', [2, 7])
def test_random_media_with_more_than_255_materials(seed):
    """The sparse list's code word carries 1 + material of the cell in its fourth byte, the per-material edge codes stop at 253: a material
    table of 306 rows (51 copies of the six rows, ids spread by position) sends cells through the id-array read and the explicit edge
    coefficients instead. Copies are the same material: every output equals the six-row run bit for bit, and the oracle within the suite's bound."""
    from babelbrain_amd import PropagationModel
    a, k = random_case(seed)
    mm, ml = a[0], a[1]
    n = ml.shape[0]
    copies = 51
    ii, jj, kk = np.meshgrid(*[np.arange(s) for s in mm.shape], indexing='ij')
    which = ((ii // 3 + 2 * (jj // 2) + 5 * kk) % copies).astype(np.uint32)
    mm2 = (mm + n * which).astype(np.uint32)
    ml2 = np.tile(ml, (copies, 1))
    if not np.isscalar(k['QCorrection']):
        k = dict(k, QCorrection=list(k['QCorrection']) * copies)       # per material
    assert ml2.shape[0] > 255 and mm2.max() >= 255 and np.isin(mm2[mm2 >= 255] % n, [1, 3, 4]).any()
    ref = PropagationModel().StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **random_case(seed)[1])
    a2 = (mm2, ml2) + tuple(a[2:])
    out = PropagationModel().StaggeredFDTD_3D_with_relaxation(*a2, SILENT=True, **k)
    compare_runs(out, ref, 0.0, both=(k['SelRMSorPeak'] == 3))
    orf = O.StaggeredFDTD_3D_with_relaxation(*a2, **k)
    compare_runs(out, orf, 1e-5, both=(k['SelRMSorPeak'] == 3))

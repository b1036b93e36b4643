"""The three-call Step-2 flow of a phased array on the device: forward run, point-source
back-propagation with on-device plane spectrum, Rayleigh re-phasing on the device, refocused forward run
(BabelIntegrationBASE.py:2338-2429, BabelIntegrationCONCAVE_PHASEDARRAY.py:395-484)."""
import numpy as np
import pytest

from babelbrain_amd import harness as H
from tests.util import oracle_dt

pytestmark = pytest.mark.gpu


def _setup():
    N1, N2, N3 = 72, 72, 104
    a, k, info = H.make_problem('C2', N=(N1, N2, N3), steps=None, stable_dt_fn=oracle_dt)
    mm, ml, f, _, _, h, T, sensor = a
    zsrc = info['zsrc']
    # array: every sub-source of a small bowl is its own element
    focal, ap = 18e-3, 16e-3
    pts, ds = H._bowl_points(focal, ap, 5, 0.0)
    depth = pts[:, 2].max()
    XDim = (np.arange(N1) - (N1 - 1) / 2) * h
    YDim = (np.arange(N2) - (N2 - 1) / 2) * h
    ZDim = depth + 2 * h + (np.arange(N3) - zsrc) * h
    Tx = {'center': pts.astype(np.float32), 'ds': ds.reshape(-1, 1).astype(np.float32), 'elemcenter': pts.astype(np.float32),
          'NumberElems': len(ds), 'elemdims': 1}
    tgt = (N1 // 2 + 3, N2 // 2 - 2, int(np.argmin(np.abs(ZDim - focal))))          # slightly off axis: phases must steer
    assert mm[tgt] == 2, 'target should sit in the brain'
    return a, k, info, Tx, (XDim, YDim, ZDim), tgt


def _three_calls(solve, forward, spectrum_of):
    """The reference's Step-2 sequence for a phased array (BASE:2338-2429): forward run, point-source run recorded
    on the entry plane, Rayleigh re-phasing (CONCAVE:407-484), forward run from the re-phased plane.
    solve(MaterialMap, ..., SensorMap, **kw) -> solver tuple; forward = ForwardSimple-compatible;
    spectrum_of(out, N1, N2, NDelta) -> complex (N1,N2) plane spectrum of call 2."""
    from babelbrain_amd import refocus
    a, k, info, Tx, (XDim, YDim, ZDim), tgt = _setup()
    mm, ml, f, _, _, h, T, sensor = a
    N1, N2, N3 = mm.shape
    zsrc, pml, dt = info['zsrc'], H.PML_THICKNESS, info['dt']
    kw = np.array(2 * np.pi * f / 1500.0 + 0j).astype(np.complex64)
    X, Y = np.meshgrid(XDim, YDim, indexing='ij')
    rf = np.stack([X.ravel(), Y.ravel(), np.full(X.size, ZDim[zsrc])], 1).astype(np.float32)
    plane = np.asarray(forward(kw, Tx['center'], Tx['ds'], np.ones(len(Tx['ds']), np.complex64), rf)).reshape(N1, N2)
    plane = refocus._clear_layer(plane, pml)
    smap, pulse = H.pulse_sources(plane, f, dt, T, N3, zsrc)
    out1 = solve(mm, ml, f, smap, pulse, h, T, sensor, **k)
    # call 2: point stress source at the target, sensors on the entry plane (BASE:2374-2398)
    _, back = H.sensor_maps(N1, N2, N3, zsrc)
    k2 = {n: v for n, v in k.items() if n not in ('Ox', 'Oy', 'Oz')}
    k2.update(TypeSource=2, SelMapsRMSPeakList=['Pressure'], SelMapsSensorsList=['Pressure'], SelRMSorPeak=1)
    out2 = solve(mm, ml, f, H.punctual_source_map(N1, N2, N3, tgt), H.punctual_source(f, dt, T), h, T, back, **k2)
    spec = spectrum_of(out2, N1, N2, pml, f, dt * k['SensorSubSampling'])
    plane2, prog = refocus.back_propagation_rayleigh(plane, spec, XDim, YDim, ZDim, zsrc, h, f, Tx, 1.0, pml, forward=forward)
    pulse2 = refocus.refocus_sources(plane, plane2, f, dt, T)
    assert pulse2.shape == pulse.shape
    out3 = solve(mm, ml, f, smap, pulse2, h, T, sensor, **k)
    return dict(plane=plane, spec=spec, plane2=plane2, prog=prog, p1=out1[2]['Pressure'], p3=out3[2]['Pressure'], tgt=tgt,
                pml=pml, zsrc=zsrc)


def _fft_plane_spectrum(out, N1, N2, pml, f, dt_sensor):
    """The plane part of CalculatePhaseData on the host, as the reference does it (BASE:2498-2499, 2523-2538):
    FFT of the sensor series, bin closest to f, raw (no 2/nTs factor)."""
    series = out[0]['Pressure']
    nTs = series.shape[1]
    b = int(np.argmin(np.abs(np.fft.fftfreq(nTs, dt_sensor) - f)))
    F = np.fft.fft(series.astype(np.float64), axis=1)[:, b]
    i, j, kk = H.decode_sensor_index(out[-1]['IndexSensorMap'], N1, N2)
    assert np.all(kk == pml)
    plane = np.zeros((N1, N2), np.complex128)
    plane[i, j] = F
    return plane


def _device_plane_spectrum(out, N1, N2, pml, f, dt_sensor):
    inp = out[-1]
    i, j, kk = H.decode_sensor_index(inp['IndexSensorMap'], N1, N2)
    assert np.all(kk == pml)
    plane = np.zeros((N1, N2), np.complex64)
    plane[i, j] = inp['SensorDFT']['Pressure'] * (out[0]['time'].size / 2.0)
    return plane


def test_refocusing_flow_through_skull():
    """Device flow (HIP solver, device Rayleigh sums, on-device DFT) against the same three calls made with the CPU
    oracles (FDTD oracle, float64 Rayleigh sum, host FFT): every intermediate and the final map agree to 1e-5."""
    from babelbrain_amd import PropagationModel, RayleighAndBHTE as R
    from oracle import oracle as O, rayleigh_oracle as RO
    model = PropagationModel()
    dev = _three_calls(lambda *a, **k: model.StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, ReturnSensorDFT=True, **k),
                       R.ForwardSimple, _device_plane_spectrum)
    ref = _three_calls(lambda *a, **k: O.StaggeredFDTD_3D_with_relaxation(*a, **k), RO.ForwardSimple, _fft_plane_spectrum)
    pml = dev['pml']
    assert np.abs(dev['spec'][pml:-pml, pml:-pml]).max() > 0 and np.all(dev['spec'][:pml, :] == 0)
    def cerr(a, b):      # rel L2 of complex (or real) arrays
        a, b = np.asarray(a, np.complex128), np.asarray(b, np.complex128)
        return float(np.sqrt(np.sum(np.abs(a - b) ** 2) / np.sum(np.abs(b) ** 2)))
    errs = {n: cerr(dev[n], ref[n]) for n in ('plane', 'spec', 'prog', 'plane2', 'p1', 'p3')}
    print('three-call flow, device vs oracles (rel L2):', {n: '%.2e' % e for n, e in errs.items()})
    for n, e in errs.items():
        assert e <= 1e-5, (n, e)
    # conjugation sign: the programming vector must be the conjugate of the field the point source sends to the elements
    assert np.allclose(np.angle(dev['prog']), np.angle(ref['prog']), atol=1e-4)
    # and the re-phased array focuses better on the (off-axis) target than the geometric one
    p1, p3, tgt, zsrc = dev['p1'], dev['p3'], dev['tgt'], dev['zsrc']
    g1 = p1[tgt] / np.sqrt(np.mean(p1[pml:-pml, pml:-pml, zsrc + 2] ** 2))
    g3 = p3[tgt] / np.sqrt(np.mean(p3[pml:-pml, pml:-pml, zsrc + 2] ** 2))
    print('focusing gain at the target: geometric %.3f, refocused %.3f' % (g1, g3))
    assert p3[tgt] > 0 and g3 > g1

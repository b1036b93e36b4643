"""Refocusing loop of the phased-array integrations, restated around the device kernels.

The reference's Step 2 for a phased array makes three solver calls (BabelIntegrationBASE.py:2372-2429):
  1. forward run from the Rayleigh source plane,
  2. a point stress source at the target, recorded on the entry plane (k = PML)           BASE:2374-2398
  3. the forward run again with the source plane rebuilt from conjugated element phases     BASE:2401-2428
Between 2 and 3 it takes the single-frequency content of the plane sensors (CalculatePhaseData,
BASE:2523-2538), back-propagates it to the element centres with the Rayleigh integral, conjugates the
phases and forward-propagates the re-phased array to the source plane
(BabelIntegrationCONCAVE_PHASEDARRAY.py:407-484). Here the Rayleigh sums run on the GPU
(RayleighAndBHTE.ForwardSimple = bfd_rayleigh_forward) and the plane spectrum comes from the on-device DFT
(bfd_get_sensor_dft); the host arithmetic in between is pinned to the reference by golden vectors
(tests/test_golden_harness.py::test_refocusing_orchestration).
"""
import numpy as np

from . import harness as H


def back_propagation_rayleigh(SourceMapRayleigh, PressMapFourierBack, XDim, YDim, ZDim, ZSourceLocation, SpatialStep,
                              Frequency, Tx, SourceAmpPa, PMLThickness, forward, c_water=1500.0, weights=1.0):
    """CONCAVE:407-454. Returns (SourceMapRayleighRefocus (N1,N2) complex, programming (nElem,) complex64).
    forward: ForwardSimple-compatible callable."""
    assert np.all(np.array(SourceMapRayleigh.shape) == np.array(PressMapFourierBack.shape))
    sel = np.abs(SourceMapRayleigh) > 0
    ypp, xpp = np.meshgrid(YDim, XDim)
    center = np.zeros((int(sel.sum()), 3), np.float32)
    center[:, 0] = xpp[sel].flatten()
    center[:, 1] = ypp[sel].flatten()
    center[:, 2] = ZDim[ZSourceLocation]
    ds = np.ones(center.shape[0]) * SpatialStep ** 2
    u0 = PressMapFourierBack[sel]
    k = np.array(2 * np.pi * Frequency / c_water + 1j * 0).astype(np.complex64)
    u2back = np.asarray(forward(k, center.astype(np.float32), ds.astype(np.float32), u0, Tx['elemcenter'].astype(np.float32)))
    nElem, edims = int(Tx['NumberElems']), int(Tx['elemdims'])
    prog = np.zeros(nElem, np.complex64)
    u0n = np.zeros((Tx['center'].shape[0], 1), np.complex64)
    for n in range(nElem):
        phi = np.angle(np.conjugate(u2back[n]))
        prog[n] = np.conjugate(u2back[n])
        u0n[n * edims:(n + 1) * edims] = (SourceAmpPa * np.exp(1j * phi)).astype(np.complex64)
    u0n = u0n * weights
    yp, xp, zp = np.meshgrid(YDim, XDim, ZDim[ZSourceLocation:ZSourceLocation + 1])      # only the source plane is kept
    rf = np.hstack((xp.reshape(-1, 1), yp.reshape(-1, 1), zp.reshape(-1, 1))).astype(np.float32)
    u2 = np.asarray(forward(k, Tx['center'].astype(np.float32), Tx['ds'].astype(np.float32), u0n, rf)).reshape(xp.shape)[:, :, 0]
    plane = u2.copy()
    p = PMLThickness
    plane[:p, :] = 0
    plane[-p:, :] = 0
    plane[:, :p] = 0
    plane[:, -p:] = 0
    return plane, prog


def refocus_sources(SourceMapRayleigh, SourceMapRayleighRefocus, freq, dt, T, ramp_length=4):
    """CreateSourcesRefocus (CONCAVE:457-484): rows follow the voxels of the ORIGINAL source mask."""
    length = np.floor(T / (1.0 / freq)) * 1 / freq
    tv = np.arange(0, length + dt, dt)
    rp = int(np.round(ramp_length / freq / dt))
    ramp = (-np.cos(np.arange(0, np.pi, np.pi / rp)) + 1) * 0.5
    ii, jj = np.where(np.abs(SourceMapRayleigh) > 0)
    u = SourceMapRayleighRefocus[ii, jj]
    n_rows = int(np.sum(np.abs(SourceMapRayleighRefocus) > 0))
    pulse = np.zeros((n_rows, tv.shape[0]))
    rows = np.abs(u)[:, None] * np.sin(2 * np.pi * freq * tv[None, :] + np.angle(u)[:, None])
    nr = min(len(ramp), tv.shape[0])
    rows[:, :nr] *= ramp[None, :nr]
    pulse[:len(ii)] = rows[:n_rows]
    return pulse


def plane_spectrum(model, args, kwargs, SensorMapBack, PunctualSource, SourceMapPunctual):
    """Call 2 (BASE:2374-2398) + the plane part of CalculatePhaseData (BASE:2523-2538): returns the
    (N1,N2) complex single-frequency field on the entry plane k = NDelta, from the on-device DFT."""
    mm, ml, f, _, _, h, T, _ = args
    kw = {k: v for k, v in kwargs.items() if k not in ('Ox', 'Oy', 'Oz')}
    kw.update(TypeSource=2, SelMapsRMSPeakList=['Pressure'], SelMapsSensorsList=['Pressure'], SelRMSorPeak=1)
    out = model.StaggeredFDTD_3D_with_relaxation(mm, ml, f, SourceMapPunctual, PunctualSource, h, T, SensorMapBack,
                                                 SILENT=True, ReturnSensorDFT=True, **kw)
    inp = out[-1]
    N1, N2, N3 = mm.shape
    i, j, k = H.decode_sensor_index(inp['IndexSensorMap'], N1, N2)
    assert np.all(k == kwargs.get('NDelta', 12))          # asserted by the reference too, BASE:2537
    plane = np.zeros((N1, N2), np.complex64)
    # CalculatePhaseData stores the raw FFT bin for the back plane (no 2/nTs factor, BASE:2533-2538)
    nTs = out[0]['time'].size
    plane[i, j] = inp['SensorDFT']['Pressure'] * (nTs / 2.0)
    return plane

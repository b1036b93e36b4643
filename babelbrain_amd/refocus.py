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


def _clear_layer(plane, width):
    """Zero the absorbing-layer frame of a source plane (nothing is injected inside the layer)."""
    out = np.array(plane, copy=True)
    frame = np.ones(out.shape, bool)
    frame[width:out.shape[0] - width, width:out.shape[1] - width] = False
    out[frame] = 0
    return out


def element_drive(field_at_elements, n_elems, subsources_per_elem, amplitude):
    """Time-reversal programming of the array: every element is driven with the conjugate phase of the field that
    reached its centre; all sub-sources of an element share the element's drive. Returns (programming (nElem,),
    per-sub-source drive (nElem*subsources_per_elem, 1)), both complex64."""
    conj = np.conjugate(np.asarray(field_at_elements).reshape(-1)[:n_elems])
    unit = np.exp(1j * np.angle(conj))        # in the precision the Rayleigh sum returned
    drive = np.repeat(amplitude * unit, subsources_per_elem).astype(np.complex64).reshape(-1, 1)
    return conj.astype(np.complex64), drive


def back_propagation_rayleigh(SourceMapRayleigh, PressMapFourierBack, XDim, YDim, ZDim, ZSourceLocation, SpatialStep,
                              Frequency, Tx, SourceAmpPa, PMLThickness, forward, c_water=1500.0, weights=1.0):
    """What BackPropagationRayleigh computes (BabelIntegrationCONCAVE_PHASEDARRAY.py:407-454), organised as two
    Rayleigh sums around `element_drive`:
      plane voxels (each a dx*dx piston carrying the plane spectrum of call 2) -> element centres,
      re-phased sub-sources of the array -> the source plane only (the reference evaluates the whole volume and keeps
      this one plane).
    Returns (SourceMapRayleighRefocus (N1,N2) complex, programming (nElem,) complex64). The outputs are held to the
    reference's own by tests/test_golden_harness.py::test_refocusing_orchestration.
    forward: ForwardSimple-compatible callable (the device kernel in production)."""
    plane_in = np.asarray(SourceMapRayleigh)
    spectrum = np.asarray(PressMapFourierBack)
    if plane_in.shape != spectrum.shape:
        raise ValueError('source plane and plane spectrum must have the same shape')
    XDim, YDim, ZDim = (np.asarray(v, np.float64) for v in (XDim, YDim, ZDim))
    z_plane = ZDim[ZSourceLocation]
    wavenumber = np.complex64(2 * np.pi * Frequency / c_water)
    wavenumber = np.array(wavenumber)

    # 1) pistons of the active plane voxels, in the row-major order boolean indexing gives
    ii, jj = np.nonzero(np.abs(plane_in) > 0)
    pistons = np.column_stack([XDim[ii], YDim[jj], np.full(ii.size, z_plane)]).astype(np.float32)
    areas = np.full(ii.size, SpatialStep ** 2, np.float32)
    at_elements = np.asarray(forward(wavenumber, pistons, areas, spectrum[ii, jj], np.asarray(Tx['elemcenter'], np.float32)))

    # 2) conjugate phases per element, spread over the element's sub-sources, optional amplitude weights
    programming, drive = element_drive(at_elements, int(Tx['NumberElems']), int(Tx['elemdims']), SourceAmpPa)
    n_sub = np.asarray(Tx['center']).shape[0]
    full = np.zeros((n_sub, 1), np.complex64)
    full[:drive.shape[0]] = drive[:n_sub]
    full = full * weights

    # 3) the re-phased array radiates onto the source plane
    gx, gy = np.meshgrid(XDim, YDim, indexing='ij')
    targets = np.column_stack([gx.ravel(), gy.ravel(), np.full(gx.size, z_plane)]).astype(np.float32)
    on_plane = np.asarray(forward(wavenumber, np.asarray(Tx['center'], np.float32), np.asarray(Tx['ds'], np.float32), full, targets))
    return _clear_layer(on_plane.reshape(gx.shape), PMLThickness), programming


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

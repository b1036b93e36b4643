"""What the caller does with the solver's maps after the three calls: crop, zero, flip, `DataForSim`.

Host-side restatement of `SimulationConditionsBASE.ReturnResults` (BASE:2729-2896) on plain arrays, so
that a pipeline built on this package (device DFT maps from `PropagationModel(..., ReturnSensorDFT=True)`
/ `harness.phase_maps`) can emit the `*DataForSim.h5` file that Step 3 (`CalculateTemperatureEffects.py:684-743`)
and the GUI (`_BabelBaseTx.py:134-174`) read, through `datafile.SaveToH5py`. Held to the reference's own
method by `tests/test_golden_harness.py` (vectors `rr_*` from `tests/golden/make_golden.py`).

Conventions kept from the reference: the simulation domain is (N1,N2,N3) including absorbing layer and
padding; `[XLOffset:-XROffset, YLOffset:-YROffset, ZLOffset:-ZROffset]` is the part that is kept; every
field is zeroed for k <= ZSourceLocation; volumes are flipped along Z on the way out; `TargetLocation` is
the focal voxel in the cropped, not yet flipped, index space.
"""
import numpy as np


class Crop:
    """Offsets of the kept region in the simulation domain and where it lands in the mask volume."""

    def __init__(self, XLOffset, XROffset, YLOffset, YROffset, ZLOffset, ZROffset, XShrink_L=0, YShrink_L=0, ZShrink_L=0):
        for v in (XROffset, YROffset, ZROffset):
            if v <= 0:
                raise ValueError('right offsets must be >= 1 (the reference slices with -offset, BASE:2749-2751)')
        self.lo = (int(XLOffset), int(YLOffset), int(ZLOffset))
        self.hi = (int(XROffset), int(YROffset), int(ZROffset))
        self.shrink = (int(XShrink_L), int(YShrink_L), int(ZShrink_L))

    def inner(self, a):
        (x0, y0, z0), (x1, y1, z1) = self.lo, self.hi
        return a[x0:-x1, y0:-y1, z0:-z1]

    def inner_xy(self, a):
        (x0, y0, _), (x1, y1, _) = self.lo, self.hi
        return a[x0:-x1, y0:-y1]


def zero_up_to_source(a, ZSourceLocation):
    """BASE:2746, 2767-2769: nothing at or before the source plane is reported. In place, like the reference."""
    a[:, :, :ZSourceLocation + 1] = 0
    return a


def paste_and_flip(field, crop, orig_shape, dtype):
    """BASE:2772-2779: kept region pasted into a zero volume of the mask's shape, then flipped along Z."""
    out = np.zeros(orig_shape, dtype)
    sub = crop.inner(field)
    sx, sy, sz = crop.shrink
    out[sx:sx + sub.shape[0], sy:sy + sub.shape[1], sz:sz + sub.shape[2]] = sub
    return np.flip(out, axis=2)


def full_solution_maps(peak, phase, crop, orig_shape, ZSourceLocation):
    """-> (FullSolutionPressure, FullSolutionPhase, MaskCalcRegions), BASE:2767-2792."""
    zero_up_to_source(peak, ZSourceLocation)
    zero_up_to_source(phase, ZSourceLocation)
    mask = np.zeros(orig_shape, bool)
    sub = crop.inner(peak).shape
    sx, sy, sz = crop.shrink
    mask[sx:sx + sub[0], sy:sy + sub[1], sz:sz + sub[2]] = True
    return (paste_and_flip(peak, crop, orig_shape, np.float32), paste_and_flip(phase, crop, orig_shape, np.float32),
            np.flip(mask, axis=2))


def rayleigh_water_maps(u2, crop, orig_shape, ZSourceLocation, skull_mask):
    """-> (RayleighWater amplitude, overlay with the mask, phase), BASE:2746-2765."""
    zero_up_to_source(u2, ZSourceLocation)
    w = paste_and_flip(u2, crop, orig_shape, np.complex64)
    amp = np.abs(w)
    overlay = amp + np.flip(skull_mask.astype(np.float32), axis=2) * amp.max() / 10
    return amp, overlay, np.angle(w)


def data_for_sim(crop, ZSourceLocation, InPeakValue, PressMapFourier, MaterialMap, FocalSpotLocation, Material,
                 XDim, YDim, ZDim, SpatialStep, zLengthBeyonFocalPoint, PMLThickness=None, SourceMapRayleigh=None,
                 InPeakValueRefocus=None, PressMapFourierRefocus=None, PressMapFourierBack=None,
                 MaterialMapCT=None, AirMask=None, ExtraMaps=None, RayleighWaterField=None):
    """The `DataForSim` dictionary of BASE:2812-2885.

    InPeakValue (f32) / PressMapFourier (c64): call 1 maps on the full domain (the caller already applied
    `Correction*sqrt(2)`, BASE:2439-2440). MaterialMap: the uint32 map (without CT ids; pass the CT one as
    MaterialMapCT). Refocus / Back arguments present <=> bDoRefocusing. ExtraMaps: other entries of
    `_DictPeakValue` when stresses or displacements are saved (BASE:2815-2820). RayleighWaterField <=>
    bUseRayleighForWater (BASE:2866-2870)."""
    d = {}
    zs = ZSourceLocation
    d['p_amp'] = crop.inner(zero_up_to_source(InPeakValue, zs)).copy()
    for k, v in (ExtraMaps or {}).items():
        if k != 'Pressure':
            d[k] = crop.inner(v)
    d['p_complex'] = crop.inner(zero_up_to_source(PressMapFourier, zs)).copy()
    unflipped = set()
    if InPeakValueRefocus is not None:
        d['p_amp_refocus'] = crop.inner(zero_up_to_source(InPeakValueRefocus, zs)).copy()
        d['p_complex_refocus'] = crop.inner(zero_up_to_source(PressMapFourierRefocus, zs)).copy()
        if PressMapFourierBack.ndim == 3:
            d['p_complex_back'] = crop.inner(PressMapFourierBack).copy()
        else:
            d['p_complex_back'] = crop.inner_xy(PressMapFourierBack).copy()     # the sensor plane: not flipped (BASE:2873)
            unflipped.add('p_complex_back')
    if MaterialMapCT is not None:
        d['MaterialMapCT'] = crop.inner(MaterialMapCT).copy()
    mm = crop.inner(MaterialMap).copy()
    d['MaterialMap'] = mm
    if AirMask is not None:
        d['AirMask'] = crop.inner(AirMask).astype(np.uint8)
    f = np.asarray(FocalSpotLocation, int) - np.array(crop.lo)
    if np.any(f < 0) or np.any(f >= np.array(mm.shape)):
        raise ValueError('the focal spot lies outside the kept region')
    if RayleighWaterField is not None:
        d['p_complex_water'] = crop.inner(RayleighWaterField)
        d['p_amp_water'] = np.abs(d['p_complex_water'])
    for k in d:
        if k not in unflipped:
            d[k] = np.flip(d[k], axis=2)
    d['Material'] = np.asarray(Material)
    (x0, y0, z0), (x1, y1, z1) = crop.lo, crop.hi
    d['x_vec'] = np.asarray(XDim)[x0:-x1]
    d['y_vec'] = np.asarray(YDim)[y0:-y1]
    d['z_vec'] = np.asarray(ZDim)[z0:-z1]
    d['SpatialStep'] = SpatialStep
    d['TargetLocation'] = f.astype(np.int64)
    d['zLengthBeyonFocalPoint'] = zLengthBeyonFocalPoint
    if SourceMapRayleigh is not None:
        p = int(PMLThickness)
        d['SourcePlane'] = SourceMapRayleigh[p:-p, p:-p]
    return d


def subsample_data_for_sim(d, ss, bDoRefocusing=False):
    """Step10_GetResults, BASE:1520-1536: optional coarser copy for saving."""
    if ss <= 1:
        return d
    kt = ['p_amp', 'p_complex', 'MaterialMap'] + [k for k in ('MaterialMapCT', 'AirMask') if k in d]
    if bDoRefocusing:
        kt += ['p_amp_refocus', 'p_complex_refocus']
    for k in kt:
        d[k] = d[k][::ss, ::ss, ::ss]
    for k in ('x_vec', 'y_vec', 'z_vec'):
        d[k] = d[k][::ss]
    d['SpatialStep'] = d['SpatialStep'] * ss
    d['TargetLocation'] = np.round(d['TargetLocation'] / ss).astype(int)
    return d

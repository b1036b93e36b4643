"""Re-creation of the reference's own acceptance study of its solver (test infrastructure).

/root/reference/OfflineBatchExamples/CompareRayleightWithFDTD compares, for 309 water-only cases, the field of
"Rayleigh source plane + BabelViscoFDTD" with the Rayleigh integral alone, and stores per-case metrics in
SummaryAnalysis.xlsx (parsed into tests/golden/rayleigh_study.json by tests/golden/make_rayleigh_study.py). Those are the
only numbers in the reference tree that its solver produced. This module rebuilds the Single-transducer cases the way
the reference's driver builds them and computes the notebook's metrics, so the engine can be held to the workbook row by
row:

  * domain, PART_1 cell 8 + BASE:1853-2068 (bTightNarrowBeamDomain): laterally the voxels within RadiusFace = 1.1 * Aperture/2
    of the axis, axially from the source plane to zLengthBeyonFocalPointWhenNarrow = 80 mm past the target, 12-cell
    absorbing layer around; water only (bWaterOnly: the material list is [Water], BASE:1306-1377);
  * time step: CalculateMatricesForPropagation(AlphaCFL = 0.5) snapped by the PPP rule (BASE:1799-1828); duration and
    sensor window from BASE:2082-2109;
  * source: Rayleigh integral of the bowl on the source plane (Single:250-304), sources per Single:313-346; the rim of
    the bowl sits in the source plane for ZAdj = 0, 10 mm above it for ZAdj = -10, and for ZAdj = +10 the plane itself
    moves 10 mm into the domain (PART_1 cell 8: TxMechanicalAdjustmentZ, ZIntoSkin);
  * result: RMS map * sqrt(2) * dispersion Correction (BASE:2433-2440), zeroed up to the source plane, cropped like the
    `_Sub` volumes (BASE:1459-1512); metrics of PART_2 cell 5 (`qcheck`).

What the reference tree does not tell is the depth of the study's target under the top of its mask (it comes from a
patient dataset); it only sets how far the domain extends in z (target depth + 80 mm). DEPTH_TARGET is an assumption.
"""
import numpy as np
from scipy import ndimage

from babelbrain_amd import harness as H

DEPTH_TARGET = 65e-3          # assumed distance source plane -> target voxel (unknown, see above)
Z_BEYOND = 80e-3              # zLengthBeyonFocalPointWhenNarrow of the Single runs (PART_1 cell 12)
C_WATER = 1500.0              # Material['Water'][1]


def build_case(freq, ppw, focal, diam, zadj, stable_dt_fn, forward, depth_target=DEPTH_TARGET, gap_vox=1.0, pml=None, rings=None,
               dout=None, zsteer=0.0, skin_offset=0.0, z_beyond=Z_BEYOND):
    """-> dict with the solver arguments of one water case and the Rayleigh field on the whole domain.
    Single transducer: a bowl (focal, diam). Annular array (CTX_500): rings = (InDiameters, OutDiameters) on the same bowl,
    dout = the distance out-plane -> focus the study uses for it (52.4 mm), every ring driven with the phase that makes
    its field arrive in phase at the steering point zsteer beyond the geometric focus (ANNULAR:359-420); skin_offset = how far
    the skin on the line of sight lies below the top of the study's mask (enters the cone width and the steering point)."""
    pml = H.PML_THICKNESS if pml is None else pml
    h = H.SSOS_AT_WATER_DENSITY / freq / ppw                       # GetSmallestSOS(f, bShear=True) is its floor at every study frequency
    water = np.array([H.MATERIALS[500e3]['Water']], np.float64)
    dt_ideal = stable_dt_fn(water, freq, h, H.ALPHA_CFL)
    dt_water = stable_dt_fn(water, freq, h, 1.0)
    ppp, dt = H.ppp_rule(dt_ideal, freq)
    dout_geom = np.sqrt(focal ** 2 - (diam / 2) ** 2)
    dout = dout_geom if dout is None else dout
    # BASE:1929-1944: the kept lateral region is the widest section of the beam cone, 1.1 * min(DistanceToFocus * tan(alpha),
    # Aperture/2), DistanceToFocus = source plane -> geometric focus = DOut + ZAdj (a transducer pulled back by 10 mm has its
    # focus 10 mm closer to the plane, and the cone is cut where it is narrower: the workbook's 'L Inf location' of those
    # cases sits at the centre of exactly this many voxels)
    alpha0 = np.arcsin(diam / 2 / focal)
    radius_face = 1.1 * min((dout + min(zadj, 0.0) + skin_offset) * np.tan(alpha0), diam / 2)
    n_half = int(np.floor(radius_face / h + 1e-9))
    n_lat = 2 * n_half + 1
    nz = int(np.round(depth_target / h)) + int(z_beyond / h) + 1
    N1 = N2 = n_lat + 2 * pml
    N3 = nz + 2 * pml
    # ZAdj > 0 pushes the transducer "into the skin": the source plane moves ZIntoSkin deeper into the domain (BASE:1839-1841)
    zsrc = pml + (int(np.round(zadj / h)) if zadj > 0 else 0)
    xs = (np.arange(N1) - (pml + n_half)) * h
    # z measured from the source plane; the rim of the bowl is `gap` above it (one voxel for ZAdj = 0: the reference moves
    # the transducer back by whole voxels until its sub-sources are behind the plane, Single:268-274)
    gap = gap_vox * h if zadj >= 0 else -zadj
    zs = (np.arange(N3) - zsrc) * h
    lam = 1482.0 / freq                                            # SpeedofSoundWater(20.0) ~ 1482 m/s sets the sub-source size
    k = np.array(2 * np.pi * freq / C_WATER + 0j).astype(np.complex64)
    if rings is None:
        alpha = np.arcsin(diam / 2 / focal)
        n_rings = max(int(np.ceil(alpha * focal / (lam / 5))), 4)  # GenerateFocusTx: PPWSurface = 5 (Single:132-137)
        pts, ds = H._bowl_points(focal, diam, n_rings, 0.0)        # apex at z = 0, focus at z = focal, rim at focal - dout
        pts = pts.copy()
        pts[:, 2] += -(focal - dout) - gap                         # rim plane at z = -gap
        u0 = np.ones(len(ds), np.complex64)
    else:
        # the array's geometric focus sits dout + zadj below the source plane (its housing, not its outermost ring, touches
        # the skin); sub-sources of lambda/8 (ANNULAR:131-137, PPWSurface = 8)
        ind, outd = rings
        alpha = np.arcsin(max(outd) / 2 / focal)
        n_r = max(int(np.ceil(alpha * focal / (lam / 8))), 8)
        pts, ds = H._bowl_points(focal, max(outd), n_r, 0.0)
        rho2 = np.hypot(pts[:, 0], pts[:, 1]) * 2
        ring = np.full(len(ds), -1)
        for r, (a, b) in enumerate(zip(ind, outd)):
            ring[(rho2 >= a) & (rho2 <= b)] = r
        pts, ds, ring = pts[ring >= 0].copy(), ds[ring >= 0], ring[ring >= 0]
        z_focus = dout + (zadj if zadj < 0 else 0.0)               # ZAdj > 0 moves the plane, not the array
        pts[:, 2] += z_focus - focal
        steer = np.array([[0.0, 0.0, z_focus + skin_offset + zsteer]], np.float32)
        u0 = np.zeros(len(ds), np.complex64)
        for r in range(len(ind)):                                   # phase of ring r = -angle(its own field at the steering point)
            sel = ring == r
            back = np.asarray(forward(k, pts[sel].astype(np.float32), ds[sel].astype(np.float32), np.ones(int(sel.sum()), np.complex64), steer))[0]
            u0[sel] = np.exp(-1j * np.angle(back))
    k = np.array(2 * np.pi * freq / C_WATER + 0j).astype(np.complex64)
    X, Y, Z = np.meshgrid(xs, xs, zs, indexing='ij')
    rf = np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1).astype(np.float32)
    del X, Y, Z
    u2 = np.asarray(forward(k, pts.astype(np.float32), ds.astype(np.float32), u0, rf)).reshape(N1, N2, N3)
    plane = u2[:, :, zsrc].copy()
    plane[:pml, :] = 0; plane[-pml:, :] = 0; plane[:, :pml] = 0; plane[:, -pml:] = 0
    T, nt, sub, start = H.time_plan(N1, N2, N3, h, dt, ppp, C_WATER)
    smap, pulse = H.pulse_sources(plane, freq, dt, T, N3, zsrc)
    sensor = np.zeros((N1, N2, N3), np.uint32)
    sensor[N1 // 2, N2 // 2, zsrc + 1:-pml] = 1                    # the full sensor volume is not needed for the RMS map
    rho_c = water[0, 0] * water[0, 1]
    args = (np.zeros((N1, N2, N3), np.uint32), water, freq, smap, pulse, h, T, sensor)
    kwargs = dict(Ox=np.array([0.0]), Oy=np.array([0.0]), Oz=np.array([1.0 / rho_c]), NDelta=pml, DT=dt,
                  ReflectionLimit=H.REFLECTION_LIMIT, USE_SINGLE=True, SelMapsRMSPeakList=['Pressure'],
                  SelMapsSensorsList=['Pressure'], SelRMSorPeak=1, AlphaCFL=1.0, TypeSource=0, QfactorCorrection=True,
                  QCorrection=1.0, SensorSubSampling=sub, SensorStart=start, ReflectorMask=None)
    return dict(args=args, kwargs=kwargs, u2=u2, pml=pml, h=h, dt=dt, dt_water=dt_water, ppp=ppp, nt=nt, zsrc=zsrc, N=(N1, N2, N3),
                n_sources=pulse.shape[0])


H317 = dict(focal=135e-3, aperture=160e-3, elem_diam=9.5e-3, z_beyond=40e-3,      # H317.py:58-59, PART_1 cells 20-22
            depth_target=76.5e-3)      # source plane -> target: the workbook's far-end error locations (z index 156 at 6 points per wavelength) give it
# how many voxels the skin on the line of sight lies below the top of the study's mask at the H317 resolutions (it enters
# the cone width and the depth of the steering point): fitted to the workbook's domain sizes -- its 'L Inf location' rows
# put the beam axis at lateral index 33 / 39 / 45 at 250 kHz, 6 points per wavelength, and 51 / 60 / 68 at 9
H317_SKIN_VOXELS = {(250, 6): 0, (250, 9): 2, (750, 6): 5}      # (kHz, points per wavelength); 750 kHz: axis index 103 / 218


def h317_subsources(freq, elements_json, ppw_surface=8):
    """The 128 spherical-cap elements of the H317 array as Rayleigh sub-sources (H317.py:56-125: every element is the cap
    GenerateFocusTx(f, 135 mm, 9.5 mm, c, PPWSurface=8) turned about the focus onto its centre). Apex plane at z = 0, focus at
    (0, 0, F). Returns (points (M,3), areas (M,), element index of every point, element centres (128,3))."""
    import json
    d = json.load(open(elements_json))
    F, ed = d['focal_m'], d['element_diameter_m']
    cen = np.array(d['centres_m'], np.float64)
    lam = 1482.0 / freq
    amax = np.arcsin(ed / 2 / F)
    n_rings = max(int(np.ceil(F * amax / (lam / ppw_surface))), 4)
    cap, ds = H._bowl_points(F, ed, n_rings, 0.0)                 # apex at the origin, focus at (0, 0, F)
    rel = cap - np.array([0.0, 0.0, F])                           # about the focus: the cap's axis is -z
    focus = np.array([0.0, 0.0, F])
    pts, areas, owner = [], [], []
    for e, c in enumerate(cen):
        n = (c - focus) / np.linalg.norm(c - focus)               # unit vector focus -> element centre
        # rotation taking -z to n (Rodrigues; the cap is symmetric about its axis, so any such rotation will do)
        a = np.array([0.0, 0.0, -1.0])
        v = np.cross(a, n); cth = float(np.dot(a, n))
        K = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
        Rm = np.eye(3) + K + K @ K / (1.0 + cth)
        pts.append(rel @ Rm.T + focus); areas.append(ds); owner.append(np.full(len(ds), e))
    return np.concatenate(pts), np.concatenate(areas), np.concatenate(owner), cen


def build_case_h317(freq, ppw, cone, xsteer, ysteer, zsteer, stable_dt_fn, forward, elements_json, depth_target=DEPTH_TARGET, pml=None,
                    skin_offset=None):
    """One H317 case of the study (PART_1 cells 8, 20-22; BabelIntegrationCONCAVE_PHASEDARRAY.py:142-147, 240-330): the array's
    geometric focus lies DistanceConeToFocus below the source plane (TxMechanicalAdjustmentZ = cone - skin distance puts it
    there whatever the depth of the target), the kept lateral region is 1.1 * min((cone + skin offset + max(ZSteer, 0)) *
    tan(alpha), Aperture / 2) with alpha = asin(Aperture / 2 / (F + max(ZSteer, 0))) (BASE:1929-1944), the domain ends 40 mm
    past the target; every element is driven with the conjugate phase of the field a point source at the steering location
    (focus + steering + skin offset) produces at its centre (CONCAVE:296-320)."""
    c0 = dict(H317)
    pml = H.PML_THICKNESS if pml is None else pml
    h = H.SSOS_AT_WATER_DENSITY / freq / ppw
    c0['skin_offset'] = H317_SKIN_VOXELS.get((int(round(freq / 1e3)), ppw), 1) * h if skin_offset is None else skin_offset
    water = np.array([H.MATERIALS[500e3]['Water']], np.float64)
    dt_ideal = stable_dt_fn(water, freq, h, H.ALPHA_CFL)
    dt_water = stable_dt_fn(water, freq, h, 1.0)
    ppp, dt = H.ppp_rule(dt_ideal, freq)
    extra = max(zsteer, 0.0)
    alpha = np.arcsin(c0['aperture'] / 2 / (c0['focal'] + extra))
    radius_face = 1.1 * min((cone + c0['skin_offset'] + extra) * np.tan(alpha), c0['aperture'] / 2)
    n_half = int(np.floor(radius_face / h + 1e-9))
    # lateral steering: the kept region is the union of the disc around the axis and the same disc around the steered axis
    # (ExtraAdjustX / Y = the steering, CONCAVE:86-89, BASE:1982-1990), cut to its bounding box
    nxh = int(np.floor((max(xsteer, 0.0) + radius_face) / h + 1e-9)); nxl = int(np.floor((max(-xsteer, 0.0) + radius_face) / h + 1e-9))
    nyh = int(np.floor((max(ysteer, 0.0) + radius_face) / h + 1e-9)); nyl = int(np.floor((max(-ysteer, 0.0) + radius_face) / h + 1e-9))
    nz = int(np.round(depth_target / h)) + int(c0['z_beyond'] / h) + 1
    N1 = nxl + nxh + 1 + 2 * pml
    N2 = nyl + nyh + 1 + 2 * pml
    N3 = nz + 2 * pml
    zsrc = pml
    xs = (np.arange(N1) - (pml + nxl)) * h
    ys = (np.arange(N2) - (pml + nyl)) * h
    zs = (np.arange(N3) - zsrc) * h
    pts, ds, owner, cen = h317_subsources(freq, elements_json)
    shift = np.array([0.0, 0.0, cone - c0['focal']])              # geometric focus at z = cone below the source plane
    pts = pts + shift; cen = cen + shift
    # whole voxels back until every sub-source is behind the plane (CONCAVE:268-274)
    while pts[:, 2].max() >= 0.0:
        pts[:, 2] -= h; cen[:, 2] -= h
    k = np.array(2 * np.pi * freq / C_WATER + 0j).astype(np.complex64)
    u0 = np.ones(len(ds), np.complex64)
    if xsteer != 0.0 or ysteer != 0.0 or zsteer != 0.0:
        steer = np.array([[xsteer, ysteer, cone + c0['skin_offset'] + zsteer]], np.float32)
        back = np.asarray(forward(k, steer, np.array([h * h], np.float32), np.ones(1, np.complex64), cen.astype(np.float32)))
        u0 = np.exp(1j * np.angle(np.conjugate(back)))[owner].astype(np.complex64)
    X, Y, Z = np.meshgrid(xs, ys, zs, indexing='ij')
    rf = np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1).astype(np.float32)
    del X, Y, Z
    u2 = np.asarray(forward(k, pts.astype(np.float32), ds.astype(np.float32), u0, rf)).reshape(N1, N2, N3)
    plane = u2[:, :, zsrc].copy()
    plane[:pml, :] = 0; plane[-pml:, :] = 0; plane[:, :pml] = 0; plane[:, -pml:] = 0
    T, nt, sub, start = H.time_plan(N1, N2, N3, h, dt, ppp, C_WATER)
    smap, pulse = H.pulse_sources(plane, freq, dt, T, N3, zsrc)
    sensor = np.zeros((N1, N2, N3), np.uint32)
    sensor[N1 // 2, N2 // 2, zsrc + 1:-pml] = 1
    rho_c = water[0, 0] * water[0, 1]
    args = (np.zeros((N1, N2, N3), np.uint32), water, freq, smap, pulse, h, T, sensor)
    kwargs = dict(Ox=np.array([0.0]), Oy=np.array([0.0]), Oz=np.array([1.0 / rho_c]), NDelta=pml, DT=dt,
                  ReflectionLimit=H.REFLECTION_LIMIT, USE_SINGLE=True, SelMapsRMSPeakList=['Pressure'],
                  SelMapsSensorsList=['Pressure'], SelRMSorPeak=1, AlphaCFL=1.0, TypeSource=0, QfactorCorrection=True,
                  QCorrection=1.0, SensorSubSampling=sub, SensorStart=start, ReflectorMask=None)
    return dict(args=args, kwargs=kwargs, u2=u2, pml=pml, h=h, dt=dt, dt_water=dt_water, ppp=ppp, nt=nt, zsrc=zsrc, N=(N1, N2, N3),
                n_sources=pulse.shape[0])


REMOPD = dict(pitch=3.08e-3, kerf=0.5e-3, aperture=0.058, z_beyond=60e-3, depth_target=78e-3)      # BabelIntegrationREMOPD.py:29-34, PART_1 cell 24


def build_case_remopd(freq, ppw, xsteer, ysteer, zsteer, stable_dt_fn, forward, elements_json, depth_target=None, pml=None):
    """One REMOPD case of the study (PART_1 cells 24-26; BabelIntegrationREMOPD.py:40-86, 280-370): a flat 16 x 16 array of
    2.58 mm square elements (pitch 3.08 mm), each sampled at about lambda(1500 m/s) / 12, one voxel behind the source plane;
    every element driven with the conjugate phase of a point source at (XSteer, YSteer, ZSteer below the source plane); kept
    region: the discs of radius 1.1 * Aperture / 2 around the axis and around the steered axis (FocalLength = 0 branch of
    BASE:1946-1951, ExtraAdjust BASE:1982-1990); the domain ends 60 mm past the target."""
    import json
    c0 = REMOPD
    pml = H.PML_THICKNESS if pml is None else pml
    h = H.SSOS_AT_WATER_DENSITY / freq / ppw
    water = np.array([H.MATERIALS[500e3]['Water']], np.float64)
    dt_ideal = stable_dt_fn(water, freq, h, H.ALPHA_CFL)
    dt_water = stable_dt_fn(water, freq, h, 1.0)
    ppp, dt = H.ppp_rule(dt_ideal, freq)
    radius_face = 1.1 * c0['aperture'] / 2
    nxh = int(np.floor((max(xsteer, 0.0) + radius_face) / h + 1e-9)); nxl = int(np.floor((max(-xsteer, 0.0) + radius_face) / h + 1e-9))
    nyh = int(np.floor((max(ysteer, 0.0) + radius_face) / h + 1e-9)); nyl = int(np.floor((max(-ysteer, 0.0) + radius_face) / h + 1e-9))
    depth = c0['depth_target'] if depth_target is None else depth_target
    nz = int(np.round(depth / h)) + int(c0['z_beyond'] / h) + 1
    N1, N2, N3 = nxl + nxh + 1 + 2 * pml, nyl + nyh + 1 + 2 * pml, nz + 2 * pml
    zsrc = pml
    xs = (np.arange(N1) - (pml + nxl)) * h
    ys = (np.arange(N2) - (pml + nyl)) * h
    zs = (np.arange(N3) - zsrc) * h
    cen = np.array(json.load(open(elements_json))['centres_m'], np.float64)
    side = c0['pitch'] - c0['kerf']
    n_lat = int(np.round(side / (1500.0 / freq / 12.0)))
    step = side / n_lat
    cx = np.arange(n_lat) * step
    cx -= cx.mean()
    gx, gy = np.meshgrid(cx, cx)
    sub = np.stack([gx.ravel(), gy.ravel(), np.zeros(gx.size)], 1)
    pts = (cen[:, None, :] + sub[None, :, :]).reshape(-1, 3)
    owner = np.repeat(np.arange(len(cen)), len(sub))
    ds = np.full(len(pts), step * step)
    pts[:, 2] = -h; cen = cen.copy(); cen[:, 2] = -h          # one voxel behind the source plane (REMOPD.py:293)
    k = np.array(2 * np.pi * freq / C_WATER + 0j).astype(np.complex64)
    u0 = np.ones(len(ds), np.complex64)
    if xsteer != 0.0 or ysteer != 0.0 or zsteer != 0.0:
        steer = np.array([[xsteer, ysteer, zsteer]], np.float32)
        back = np.asarray(forward(k, steer, np.array([h * h], np.float32), np.ones(1, np.complex64), cen.astype(np.float32)))
        u0 = np.exp(1j * np.angle(np.conjugate(back)))[owner].astype(np.complex64)
    X, Y, Z = np.meshgrid(xs, ys, zs, indexing='ij')
    rf = np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1).astype(np.float32)
    del X, Y, Z
    u2 = np.asarray(forward(k, pts.astype(np.float32), ds.astype(np.float32), u0, rf)).reshape(N1, N2, N3)
    plane = u2[:, :, zsrc].copy()
    import os
    if 'REMOPD_PLANE_SHIFT' in os.environ:            # experiment: source values taken a fraction of a voxel off the plane
        Xp, Yp = np.meshgrid(xs, ys, indexing='ij')
        rp = np.stack([Xp.ravel(), Yp.ravel(), np.full(Xp.size, float(os.environ['REMOPD_PLANE_SHIFT']) * h)], 1).astype(np.float32)
        plane = np.asarray(forward(k, pts.astype(np.float32), ds.astype(np.float32), u0, rp)).reshape(N1, N2)
    plane[:pml, :] = 0; plane[-pml:, :] = 0; plane[:, :pml] = 0; plane[:, -pml:] = 0
    T, nt, sub_s, start = H.time_plan(N1, N2, N3, h, dt, ppp, C_WATER)
    smap, pulse = H.pulse_sources(plane, freq, dt, T, N3, zsrc)
    sensor = np.zeros((N1, N2, N3), np.uint32)
    sensor[N1 // 2, N2 // 2, zsrc + 1:-pml] = 1
    rho_c = water[0, 0] * water[0, 1]
    args = (np.zeros((N1, N2, N3), np.uint32), water, freq, smap, pulse, h, T, sensor)
    kwargs = dict(Ox=np.array([0.0]), Oy=np.array([0.0]), Oz=np.array([1.0 / rho_c]), NDelta=pml, DT=dt,
                  ReflectionLimit=H.REFLECTION_LIMIT, USE_SINGLE=True, SelMapsRMSPeakList=['Pressure'],
                  SelMapsSensorsList=['Pressure'], SelRMSorPeak=1, AlphaCFL=1.0, TypeSource=0, QfactorCorrection=True,
                  QCorrection=1.0, SensorSubSampling=sub_s, SensorStart=start, ReflectorMask=None)
    return dict(args=args, kwargs=kwargs, u2=u2, pml=pml, h=h, dt=dt, dt_water=dt_water, ppp=ppp, nt=nt, zsrc=zsrc, N=(N1, N2, N3),
                n_sources=pulse.shape[0], water_gap_voxels=int(np.round(1.2e-3 / h)))       # ZTxCorrecton = 1.2 mm (REMOPD.py:34, 272)


def result_volumes(case, rms_pressure):
    """FDTD and Rayleigh amplitude volumes as the study compares them: Correction * sqrt(2) (BASE:2433-2440), zero up to the
    source plane (BASE:2746, 2767-2769), the `_Sub` crop (interior without its last row / column and without the source
    plane, BASE:1488-1512 after the Z flip is undone)."""
    pml, zsrc = case['pml'], case['zsrc']
    gap = case.get('water_gap_voxels', 0)      # flat arrays: ZTxCorrecton puts that many planes of water above the mask (BASE:1859)
    corr = H.dispersion_correction(case['dt'], case['dt_water'])
    A = np.array(rms_pressure, np.float64) * corr * np.sqrt(2.0)
    B = np.abs(case['u2']).astype(np.float64)
    out = []
    for v in (A, B):
        v[:, :, :zsrc + 1] = 0
        c = v[pml:-pml, pml:-pml, pml + gap:-pml]
        out.append(c[:-1, :-1, 1:])
    return out[0], out[1]


def _focal_region(data, voxel):
    """CalcVolumetricMetrics of PART_2 cell 5: largest connected region at >= half of the maximum."""
    lab, n = ndimage.label(data / data.max() >= 0.5)
    if n > 1:
        sizes = ndimage.sum(np.ones_like(lab), lab, index=np.arange(1, n + 1))
        sel = lab == (1 + int(np.argmax(sizes)))
    else:
        sel = lab == 1
    idx = np.array(np.nonzero(sel), np.float64)
    return idx.mean(axis=1) * voxel, sel.sum() * voxel ** 3


def qcheck(A, B, voxel_mm):
    """The notebook's metrics (PART_2 cell 5) for FDTD volume A against Rayleigh volume B."""
    line = A[A.shape[0] // 2, A.shape[1] // 2, :]
    first = int(np.nonzero(line != 0.0)[0][0])
    A, B = A[:, :, first + 1:], B[:, :, first + 1:]
    ca, va = _focal_region(A, voxel_mm)
    cb, vb = _focal_region(B, voxel_mm)
    linf = 100.0 * np.abs(A - B) / B.max()
    loc = np.unravel_index(int(np.argmax(linf)), linf.shape)
    return {'Distance focal centroid': float(np.linalg.norm(ca - cb)),
            'Distance focal max': float(np.linalg.norm((np.array(np.unravel_index(int(np.argmax(A)), A.shape)) -
                                                        np.array(np.unravel_index(int(np.argmax(B)), B.shape))) * voxel_mm)),
            'Difference amplitude': float(100 * (A.max() - B.max()) / B.max()),
            'Difference volume': float(100 * (va - vb) / vb),
            'L Inf': float(linf.max()), 'L Inf location': [int(x) for x in loc],
            'L2': float(100.0 * np.sqrt(np.sum((A - B) ** 2) / np.sum(B ** 2)))}


CTX500 = dict(focal=62.94e-3, diam=64.0e-3, dout=52.4e-3, z_beyond=40e-3, skin_offset=0.9e-3,      # Babel_CTX500/default.yaml, PART_1 cell 8 / 16
              rings=(np.array([0.0, 31.6988e-3, 44.2688e-3, 53.6688e-3]), np.array([31.14e-3, 43.71e-3, 53.11e-3, 60.83e-3])))


def run_case(row, solver, stable_dt_fn, forward, depth_target=DEPTH_TARGET, gap_vox=1.0, pml=None):
    """row: an entry of rayleigh_study.json (tx 'Single' or 'CTX_500'). solver(*args, **kwargs) -> the solver tuple."""
    if row['tx'] == 'REMOPD':
        import os, re
        m = re.match(r'DEEP_REMODP_(\d+)kHz_(\d+)PPW_XSteer_(-?[\d.]+)_YSteer_(-?[\d.]+)_ZSteer_(-?[\d.]+)\.nii', row['Description'])
        fk, ppw, xs_, ys_, zs_ = int(m.group(1)), int(m.group(2)), float(m.group(3)) * 1e-3, float(m.group(4)) * 1e-3, float(m.group(5)) * 1e-3
        case = build_case_remopd(fk * 1e3, ppw, xs_, ys_, zs_, stable_dt_fn, forward,
                                 os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'remopd_elements.json'),
                                 None if depth_target == DEPTH_TARGET else depth_target, pml)
    elif row['tx'] == 'H317':
        import os, re
        m = re.match(r'ConeDistance_([\d.]+)_DEEP_H317_(\d+)kHz_(\d+)PPW_XSteer_(-?[\d.]+)_YSteer_(-?[\d.]+)_ZSteer_(-?[\d.]+)\.nii', row['Description'])
        cone, fk, ppw, xs_, ys_, zs_ = (float(m.group(1)) * 1e-3, int(m.group(2)), int(m.group(3)), float(m.group(4)) * 1e-3,
                                         float(m.group(5)) * 1e-3, float(m.group(6)) * 1e-3)
        case = build_case_h317(fk * 1e3, ppw, cone, xs_, ys_, zs_, stable_dt_fn, forward,
                               os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'h317_elements.json'),
                               H317['depth_target'] if depth_target == DEPTH_TARGET else depth_target, pml,
                               skin_offset=(float(os.environ['H317_SKIN_OFFSET_MM']) * 1e-3 if 'H317_SKIN_OFFSET_MM' in os.environ else None))
    elif row['tx'] == 'CTX_500':
        import re
        m = re.match(r'ZAdj_(-?[\d.]+)_DEEP_CTX_500_500kHz_(\d+)PPW_ZSteering_(-?[\d.]+)\.nii', row['Description'])
        zadj, ppw, zsteer = float(m.group(1)) * 1e-3, int(m.group(2)), float(m.group(3)) * 1e-3
        c = CTX500
        case = build_case(500e3, ppw, c['focal'], c['diam'], zadj, stable_dt_fn, forward, depth_target, gap_vox, pml, rings=c['rings'],
                          dout=c['dout'], zsteer=zsteer, skin_offset=c['skin_offset'], z_beyond=c['z_beyond'])
    else:
        case = build_case(row['freq_khz'] * 1e3, row['ppw'], row['focal_mm'] * 1e-3, row['diam_mm'] * 1e-3, row['zadj_mm'] * 1e-3,
                          stable_dt_fn, forward, depth_target, gap_vox, pml)
    out = solver(*case['args'], **case['kwargs'])
    A, B = result_volumes(case, out[2]['Pressure'])
    m = qcheck(A, B, case['h'] * 1e3)
    m.update(N=case['N'], nt=case['nt'], ppp=case['ppp'], cfl_water=case['dt'] / case['dt_water'], n_sources=case['n_sources'])
    return m

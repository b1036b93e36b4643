"""Drop-in for the parts of `BabelViscoFDTD.tools.RayleighAndBHTE` that BabelBrain's Step 2 uses
(imports at TranscranialModeling/BabelIntegrationBASE.py:19 and BabelIntegrationSingle.py:23):

    ForwardSimple(cwvnb, center, ds, u0, rf, deviceMetal=None)     Single:295, ANNULAR:383,411, CONCAVE:307,328,425,446
    InitCuda / InitOpenCL / InitMetal(deviceName)                  BASE:918-925
    SpeedofSoundWater(T)                                           Single:243
    GenerateFocusTx(f, Foc, Diam, c, PPWSurface)                   Single:241

ForwardSimple runs on the MI355X through the C ABI (bfd_rayleigh_forward); there is no CPU fallback.
The BHTE thermal solver of the same upstream module is SURVEY.md 8f "next #4" and is not here yet.
"""
import ctypes as C

import numpy as np

from . import _engine

_device = 0
last_kernel_ms = None


def _init(deviceName=None):
    """Device selection by name substring, as the reference's Init* functions do (BASE:918-925)."""
    global _device
    devs = _engine.list_devices()
    if not devs:
        raise _engine.EngineError('no HIP device visible')
    if deviceName:
        for d, name in devs:
            if deviceName.lower() in name.lower():
                _device = d
                break
    return devs


InitCuda = InitOpenCL = InitMetal = InitHIP = _init


def ForwardSimple(cwvnb, center, ds, u0, rf, deviceMetal=None, MacOsPlatform=None, u0step=0):
    """u2[n] = (i k / 2 pi) sum_m u0[m] ds[m] exp(-i k |rf[n]-center[m]|) / |rf[n]-center[m]|.
    cwvnb: complex wavenumber (Im k < 0 attenuates: the sum uses exp(-i k R) as written); center (M,3) f32; ds (M,) or (M,1) f32;
    u0 (M,) or (M,1) complex64; rf (N,3) f32. Returns complex64 (N,)."""
    global last_kernel_ms
    lib = _engine.load_library()
    k = complex(np.asarray(cwvnb).reshape(-1)[0])
    cen = np.ascontiguousarray(center, np.float32).reshape(-1, 3)
    a = np.ascontiguousarray(ds, np.float32).reshape(-1)
    u = np.ascontiguousarray(np.asarray(u0).reshape(-1), np.complex64)
    pts = np.ascontiguousarray(rf, np.float32).reshape(-1, 3)
    if not (len(a) == len(cen) == len(u)):
        raise ValueError('center, ds and u0 must describe the same number of sources')
    out = np.zeros(len(pts), np.complex64)
    ms = C.c_double()
    rc = lib.bfd_rayleigh_forward(_device, len(cen), cen.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p),
                                  u.view(np.float32).ctypes.data_as(C.c_void_p), k.real, k.imag, len(pts),
                                  pts.ctypes.data_as(C.c_void_p), out.view(np.float32).ctypes.data_as(C.c_void_p), C.byref(ms))
    if rc != 0:
        raise _engine.EngineError('bfd_rayleigh_forward failed (rc=%d): %s' % (rc, lib.bfd_last_error().decode()))
    last_kernel_ms = ms.value
    return out


def SpeedofSoundWater(Temperature):
    """Speed of sound in pure water, m/s, Marczak (1997) fifth-order polynomial, 0-95 degC."""
    c = [2.787860e-9, -1.398845e-6, 3.287156e-4, -5.799136e-2, 5.038813, 1.402385e3]
    return np.polyval(c, Temperature)


def GenerateFocusTx(f, Foc, Diam, c, PPWSurface=4):
    """Spherical-cap radiator (apex at the origin, focus at z=+Foc) tessellated into patches of about
    lambda/PPWSurface: dict with 'center' (M,3), 'ds' (M,1), 'normal' (M,3), 'elemcenter' (1,3),
    'VertDisplay' (= centres) and the geometry scalars. Equal-area rings as in harness._bowl_points."""
    from .harness import _bowl_points
    lam = c / f
    amax = np.arcsin(min(Diam / 2 / Foc, 1.0))
    n_rings = max(int(np.ceil(Foc * amax / (lam / PPWSurface))), 4)
    pts, ds = _bowl_points(Foc, Diam, n_rings, 0.0)
    focus = np.array([[0.0, 0.0, Foc]])
    nrm = focus - pts
    nrm /= np.linalg.norm(nrm, axis=1)[:, None]
    return {'center': pts.astype(np.float32), 'ds': ds.reshape(-1, 1).astype(np.float32), 'normal': nrm.astype(np.float32),
            'elemcenter': np.zeros((1, 3), np.float32), 'VertDisplay': pts.astype(np.float32).copy(),
            'FaceDisplay': np.zeros((0, 4), np.int64), 'NumberElems': 1, 'Aperture': Diam, 'FocalLength': Foc}


# ------------------------------------------------------------------------------------------------
# BHTE: Pennes bio-heat equation + CEM43 dose on the device (bfd_bhte_run)
# ------------------------------------------------------------------------------------------------
def bhte_coefficients(MaterialList, dx, dt, DutyCycle=1.0, blood_rho=1050.0, blood_ct=3617.0):
    """Per-material float32 coefficients of the explicit scheme (documented in csrc/bfd_bhte.hip):
    cd = dt k/(rho c dx^2), cp = dt rho_b c_b w/(6e7 c) with w in mL/min/kg, and the factor qf that turns
    p^2 into the temperature increment of one ON step: qf = dt DutyCycle Absorption Attenuation/(rho^2 c_s c)."""
    rho = np.asarray(MaterialList['Density'], np.float64)
    ct = np.asarray(MaterialList['SpecificHeat'], np.float64)
    cd = dt * np.asarray(MaterialList['Conductivity'], np.float64) / (rho * ct * dx ** 2)
    cp = dt * blood_rho * blood_ct * np.asarray(MaterialList['Perfusion'], np.float64) / (6e7 * ct)
    qf = (dt * DutyCycle * np.asarray(MaterialList['Absorption'], np.float64) * np.asarray(MaterialList['Attenuation'], np.float64)
          / (rho * np.asarray(MaterialList['SoS'], np.float64)) / (rho * ct))
    if cd.max() > 1.0 / 6.0:
        raise ValueError('BHTE time step too large: dt k/(rho c dx^2) = %.3f > 1/6' % cd.max())
    return cd.astype(np.float32), cp.astype(np.float32), qf.astype(np.float32)


def BHTE(Pressure, MaterialMap, MaterialList, dx, TotalDurationSteps, nStepsOn, LocationMonitoring,
         nFactorMonitoring=1, dt=0.1, blood_rho=1050, blood_ct=3617, stableTemp=37.0, DutyCycle=1.0,
         Backend='HIP', MonitoringPointsMap=None, initT0=None, initDose=None):
    """Same call as the reference makes (CalculateTemperatureEffects.py:365-456, 960). Pressure: amplitude map
    (N1,N2,N3) Pa; MaterialMap: integer ids; MaterialList: dict of per-material arrays 'Density', 'SoS',
    'Attenuation' (Np/m), 'SpecificHeat', 'Conductivity', 'Perfusion' (mL/min/kg), 'Absorption' (fraction of the
    attenuation that heats), 'InitTemperature'. Returns (ResTemp, ResDose, MonitorSlice, Qarr[, TemperaturePoints])."""
    global last_kernel_ms
    lib = _engine.load_library()
    P = np.asarray(Pressure)
    N1, N2, N3 = P.shape
    mm = np.asarray(MaterialMap)
    nMat = len(MaterialList['Density'])
    if mm.max() >= nMat or nMat > 256:
        raise ValueError('MaterialMap ids must index MaterialList (at most 256 materials)')
    cd, cp, qf = bhte_coefficients(MaterialList, dx, dt, DutyCycle, blood_rho, blood_ct)

    def xf(a, dtype):
        return np.ascontiguousarray(np.asarray(a).transpose(2, 1, 0), dtype=dtype)
    mat = xf(mm, np.uint8)
    p32 = xf(P, np.float32)
    q = (p32 * p32) * qf[mat]                                    # float32, same operation order as the oracle
    T = xf(initT0, np.float32) if initT0 is not None else np.asarray(MaterialList['InitTemperature'], np.float32)[mat]
    T = np.ascontiguousarray(T, np.float32)
    dose = xf(initDose, np.float32) if initDose is not None else np.zeros((N3, N2, N1), np.float32)
    nSteps = int(TotalDurationSteps)
    fm = max(int(nFactorMonitoring), 1)
    slice_ok = LocationMonitoring is not None and int(LocationMonitoring) >= 0
    nS = (nSteps + fm - 1) // fm if slice_ok else 0
    mon = np.zeros((N1, N3, nS), np.float32) if slice_ok else None
    idx = pts = None
    if MonitoringPointsMap is not None:
        mp = xf(MonitoringPointsMap, np.uint32).ravel()
        lin = np.flatnonzero(mp)
        order = np.argsort(mp[lin], kind='stable')               # point ids 1..n label the rows
        idx = np.ascontiguousarray(lin[order], np.uint32)
        pts = np.zeros((len(idx), nSteps), np.float32)
    ms = C.c_double()

    def ptr(a):
        return None if a is None else a.ctypes.data_as(C.c_void_p)
    rc = lib.bfd_bhte_run(_device, N1, N2, N3, nMat, ptr(mat), ptr(cd), ptr(cp), ptr(q), ptr(T), ptr(dose), float(stableTemp),
                          float(dt), nSteps, int(nStepsOn), int(LocationMonitoring) if slice_ok else -1, fm, ptr(mon),
                          0 if idx is None else len(idx), ptr(idx), ptr(pts), C.byref(ms))
    if rc != 0:
        raise _engine.EngineError('bfd_bhte_run failed (rc=%d): %s' % (rc, lib.bfd_last_error().decode()))
    last_kernel_ms = ms.value

    def vol(a):
        return np.ascontiguousarray(a.transpose(2, 1, 0))
    out = (vol(T), vol(dose), mon if slice_ok else np.zeros((0,), np.float32), vol(q))
    if MonitoringPointsMap is not None:
        out = out + (pts,)
    return out

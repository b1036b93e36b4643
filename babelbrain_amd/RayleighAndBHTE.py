"""Drop-in for the parts of `BabelViscoFDTD.tools.RayleighAndBHTE` that BabelBrain's Step 2 uses
(imports at TranscranialModeling/BabelIntegrationBASE.py:19 and BabelIntegrationSingle.py:23):

    ForwardSimple(cwvnb, center, ds, u0, rf, deviceMetal=None)     Single:295, ANNULAR:383,411, CONCAVE:307,328,425,446
    InitCuda / InitOpenCL / InitMetal(deviceName)                  BASE:918-925
    SpeedofSoundWater(T)                                           Single:243
    GenerateFocusTx(f, Foc, Diam, c, PPWSurface)                   Single:241

ForwardSimple runs on the MI355X through the C ABI (bfd_rayleigh_forward); there is no CPU fallback.
BHTE / BHTEMultiplePressureFields (CalculateTemperatureEffects.py:365-456, 960-990) run on the device too
(bfd_bhte_run_fields: four time steps per pass, csrc/bfd_bhte.hip).
"""
import ctypes as C
import os
import threading

import numpy as np

from . import _engine

_device = 0
_devices = None          # several HIP ordinals: ForwardSimple shares its field points among them (set_devices / BABELFDTD_DEVICES)
last_kernel_ms = None


def set_devices(devices):
    """Ordinals (or 'all') of the GPUs ForwardSimple may use together: the field points are independent, so they are dealt to
    the devices in contiguous shares, one host thread per device, no exchange (shard, no collective). None or
    one ordinal = single device. The environment variable BABELFDTD_DEVICES does the same for a caller that cannot pass
    arguments (the reference selects its device by name, BASE:918-925)."""
    global _devices, _device
    if devices is None:
        _devices = None
        return
    if isinstance(devices, str):
        devices = [d for d, _ in _engine.list_devices()] if devices.strip().lower() == 'all' else [int(x) for x in devices.split(',') if x.strip()]
    devices = [int(d) for d in devices]
    _devices = devices if len(devices) > 1 else None
    if len(devices) == 1:
        _device = devices[0]


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


InitCuda = InitOpenCL = InitMetal = InitMLX = InitHIP = _init      # the reference picks one by backend (BabelBrain.py:429-439)


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
    devices = _devices
    if devices is None and os.environ.get('BABELFDTD_DEVICES'):
        set_devices(os.environ['BABELFDTD_DEVICES'])
        devices = _devices
    if devices is None or len(pts) < 64 * len(devices):
        devices = [_device]
    bounds = [len(pts) * r // len(devices) for r in range(len(devices) + 1)]
    times, errors = [0.0] * len(devices), [None] * len(devices)

    def share(r):          # ctypes releases the GIL for the call: the devices work at the same time
        n0, n1 = bounds[r], bounds[r + 1]
        if n1 <= n0:
            return
        ms = C.c_double()
        rc = lib.bfd_rayleigh_forward(devices[r], len(cen), cen.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p),
                                      u.view(np.float32).ctypes.data_as(C.c_void_p), k.real, k.imag, n1 - n0,
                                      pts[n0:n1].ctypes.data_as(C.c_void_p), out[n0:n1].view(np.float32).ctypes.data_as(C.c_void_p), C.byref(ms))
        if rc != 0:
            errors[r] = 'bfd_rayleigh_forward failed on device %d (rc=%d): %s' % (devices[r], rc, lib.bfd_last_error().decode())
        times[r] = ms.value
    if len(devices) == 1:
        share(0)
    else:
        th = [threading.Thread(target=share, args=(r,)) for r in range(len(devices))]
        for t in th:
            t.start()
        for t in th:
            t.join()
    for e in errors:
        if e:
            raise _engine.EngineError(e)
    last_kernel_ms = max(times)
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
# Heat source per voxel from the pressure amplitude. 'linear': Q = Absorption * alpha * p^2 / (rho c) (plane-wave intensity
# times the local absorption); 'exponential': the energy a plane wave loses crossing the voxel,
# Absorption * p^2 / (rho c) * (1 - exp(-2 h alpha)) / (2 h). They differ by about h*alpha (a few per cent in skull at
# 0.3-0.5 mm voxels). Which one the absent package uses is unverified; module-level default, per call
# through bhte_coefficients(source_form=...).
BHTE_SOURCE_FORM = 'linear'


def bhte_coefficients(MaterialList, dx, dt, DutyCycle=1.0, blood_rho=1050.0, blood_ct=3617.0, source_form=None):
    """Per-material float32 coefficients of the explicit scheme (documented in csrc/bfd_bhte.hip):
    cd = dt k/(rho c dx^2), cp = dt rho_b c_b w/(6e7 c) with w in mL/min/kg, and the factor qf that turns
    p^2 into the temperature increment of one ON step: qf = dt DutyCycle Absorption A/(rho^2 c_s c), A = Attenuation
    ('linear') or (1 - exp(-2 dx Attenuation))/(2 dx) ('exponential')."""
    rho = np.asarray(MaterialList['Density'], np.float64)
    ct = np.asarray(MaterialList['SpecificHeat'], np.float64)
    cd = dt * np.asarray(MaterialList['Conductivity'], np.float64) / (rho * ct * dx ** 2)
    cp = dt * blood_rho * blood_ct * np.asarray(MaterialList['Perfusion'], np.float64) / (6e7 * ct)
    att = np.asarray(MaterialList['Attenuation'], np.float64)
    form = BHTE_SOURCE_FORM if source_form is None else source_form
    if form == 'exponential':
        att = -np.expm1(-2.0 * dx * att) / (2.0 * dx)
    elif form != 'linear':
        raise ValueError("source_form must be 'linear' or 'exponential'")
    qf = (dt * DutyCycle * np.asarray(MaterialList['Absorption'], np.float64) * att
          / (rho * np.asarray(MaterialList['SoS'], np.float64)) / (rho * ct))
    if cd.max() > 1.0 / 6.0:
        raise ValueError('BHTE time step too large: dt k/(rho c dx^2) = %.3f > 1/6' % cd.max())
    return cd.astype(np.float32), cp.astype(np.float32), qf.astype(np.float32)


def field_schedule(nStepsOnOffList, TotalDurationSteps):
    """Which pressure field heats during each step: field n is on for nStepsOnOffList[n,0] steps, then nothing for
    nStepsOnOffList[n,1] steps, then field n+1 ...; the sequence repeats until TotalDurationSteps (the caller builds
    equal slots per focal spot, CalculateTemperatureEffects.py:715-736). -1 = no heating. Documented restatement:
    the package that defines it is absent from the reference tree."""
    oo = np.asarray(nStepsOnOffList, np.int64).reshape(-1, 2)
    if np.any(oo < 0) or oo.sum() <= 0:
        raise ValueError('nStepsOnOffList needs non-negative step counts and a non-empty cycle')
    cycle = np.concatenate([np.concatenate([np.full(on, n, np.int32), np.full(off, -1, np.int32)]) for n, (on, off) in enumerate(oo)])
    reps = -(-int(TotalDurationSteps) // len(cycle)) if TotalDurationSteps > 0 else 0
    return np.ascontiguousarray(np.tile(cycle, reps)[:int(TotalDurationSteps)], np.int32)


def bhte_pass_plan(sched, nFactorMonitoring=1, monitored_plane=True, steps_heating=4, steps_cooling=4):
    """The passes bfd_bhte_run_* cuts a schedule into (the rule of bhte_run_core, restated for byte accounting): S = 4 steps per pass wherever the next S steps carry the same field; else two steps, else one. Returns [(first step, length, heating)]."""
    sched = [int(v) for v in sched]
    fm = max(int(nFactorMonitoring), 1)
    out, s, n = [], 0, len(sched)
    while s < n:
        S = steps_heating if sched[s] >= 0 else steps_cooling
        ok = S >= 3 and s + S <= n and all(sched[s + j] == sched[s] for j in range(1, S))
        L = S if ok else (2 if s + 1 < n else 1)
        out.append((s, L, sched[s] >= 0 or (L == 2 and sched[s + 1] >= 0)))
        s += L
    return out


def _bhte_run(fields, sched, MaterialMap, MaterialList, dx, LocationMonitoring, nFactorMonitoring, dt, blood_rho, blood_ct,
              stableTemp, DutyCycle, MonitoringPointsMap, initT0, initDose):
    global last_kernel_ms
    lib = _engine.load_library()
    P = np.asarray(fields)
    nF, N1, N2, N3 = P.shape
    mm = np.asarray(MaterialMap)
    nMat = len(MaterialList['Density'])
    if mm.shape != (N1, N2, N3):
        raise ValueError('MaterialMap must have the shape of the pressure field(s)')
    if mm.max() >= nMat or nMat > 256:
        raise ValueError('MaterialMap ids must index MaterialList (at most 256 materials)')
    cd, cp, qf = bhte_coefficients(MaterialList, dx, dt, DutyCycle, blood_rho, blood_ct)

    # The volumes go to the device in the caller's C order (bfd_bhte_run_volumes: last axis fastest, neighbours summed axis 0
    # first like the oracle does): no transposes; the heat source q = (p p) qf[material] is computed on the device and comes back.
    mat = np.ascontiguousarray(mm, np.uint8)
    P32 = np.ascontiguousarray(P, np.float32)
    q = np.empty((nF, N1, N2, N3), np.float32)
    flags = 0
    if initT0 is not None:
        T = np.array(initT0, np.float32, order='C'); flags |= 1
        if T.shape != (N1, N2, N3):
            raise ValueError('initT0 must have the shape of the pressure field')
    else:
        T = np.empty((N1, N2, N3), np.float32)
    if initDose is not None:
        dose = np.array(initDose, np.float32, order='C'); flags |= 2
        if dose.shape != (N1, N2, N3):
            raise ValueError('initDose must have the shape of the pressure field')
    else:
        dose = np.empty((N1, N2, N3), np.float32)
    initT = np.ascontiguousarray(MaterialList['InitTemperature'], np.float32)
    nSteps = len(sched)
    fm = max(int(nFactorMonitoring), 1)
    slice_ok = LocationMonitoring is not None and int(LocationMonitoring) >= 0
    if slice_ok and int(LocationMonitoring) >= N2:
        raise ValueError('LocationMonitoring is outside the volume')
    nS = (nSteps + fm - 1) // fm if slice_ok else 0
    mon = np.zeros((N1, N3, nS), np.float32) if slice_ok else None
    idx = pts = None
    if MonitoringPointsMap is not None:
        mp = np.asarray(MonitoringPointsMap)
        if mp.shape != (N1, N2, N3):
            raise ValueError('MonitoringPointsMap must have the shape of the pressure field')
        mp = mp.ravel()
        lin = np.flatnonzero(mp)
        order = np.argsort(mp[lin], kind='stable')               # point ids 1..n label the rows
        idx = np.ascontiguousarray(lin[order], np.uint32)
        pts = np.zeros((len(idx), nSteps), np.float32)
    ms = C.c_double()
    sched = np.ascontiguousarray(sched, np.int32)
    if sched.size == 0:
        sched = np.full(1, -1, np.int32)

    def ptr(a):
        return None if a is None else a.ctypes.data_as(C.c_void_p)
    rc = lib.bfd_bhte_run_volumes(_device, N1, N2, N3, nMat, ptr(mat), ptr(cd), ptr(cp), ptr(qf), ptr(initT), nF, ptr(P32), ptr(q), ptr(T), ptr(dose),
                                  flags, float(stableTemp), float(dt), nSteps, ptr(sched), int(LocationMonitoring) if slice_ok else -1, fm, ptr(mon),
                                  0 if idx is None else len(idx), ptr(idx), ptr(pts), C.byref(ms))
    if rc != 0:
        raise _engine.EngineError('bfd_bhte_run_volumes failed (rc=%d): %s' % (rc, lib.bfd_last_error().decode()))
    last_kernel_ms = ms.value
    out = (T, dose, mon if slice_ok else np.zeros((0,), np.float32), q)
    if MonitoringPointsMap is not None:
        out = out + (pts,)
    return out


def BHTE(Pressure, MaterialMap, MaterialList, dx, TotalDurationSteps, nStepsOn, LocationMonitoring,
         nFactorMonitoring=1, dt=0.1, blood_rho=1050, blood_ct=3617, stableTemp=37.0, DutyCycle=1.0,
         Backend='HIP', MonitoringPointsMap=None, initT0=None, initDose=None):
    """Same call as the reference makes (CalculateTemperatureEffects.py:365-456, 960). Pressure: amplitude map
    (N1,N2,N3) Pa; MaterialMap: integer ids; MaterialList: dict of per-material arrays 'Density', 'SoS',
    'Attenuation' (Np/m), 'SpecificHeat', 'Conductivity', 'Perfusion' (mL/min/kg), 'Absorption' (fraction of the
    attenuation that heats), 'InitTemperature'. Returns (ResTemp, ResDose, MonitorSlice, Qarr[, TemperaturePoints])."""
    P = np.asarray(Pressure)
    if P.ndim != 3:
        raise ValueError('Pressure must be a 3-D amplitude map')
    nSteps = int(TotalDurationSteps)
    sched = np.full(nSteps, -1, np.int32)
    sched[:max(min(int(nStepsOn), nSteps), 0)] = 0
    out = _bhte_run(P[None], sched, MaterialMap, MaterialList, dx, LocationMonitoring, nFactorMonitoring, dt, blood_rho, blood_ct,
                    stableTemp, DutyCycle, MonitoringPointsMap, initT0, initDose)
    return out[:3] + (out[3][0],) + out[4:]


def BHTEMultiplePressureFields(PressureFields, MaterialMap, MaterialList, dx, TotalDurationSteps, nStepsOnOffList, LocationMonitoring,
                               nFactorMonitoring=1, dt=0.1, blood_rho=1050, blood_ct=3617, stableTemp=37.0, Backend='HIP',
                               MonitoringPointsMap=None, initT0=None, initDose=None):
    """Same call as the reference makes for steered multi-point sonications (CalculateTemperatureEffects.py:381-394,
    978-990): PressureFields (nFields,N1,N2,N3) Pa, nStepsOnOffList (nFields,2) int = steps on / off per field (see
    `field_schedule`). No DutyCycle argument: the on/off schedule is the duty cycle. Qarr comes back per field."""
    P = np.asarray(PressureFields)
    if P.ndim != 4:
        raise ValueError('PressureFields must be (nFields, N1, N2, N3)')
    oo = np.asarray(nStepsOnOffList).reshape(-1, 2)
    if oo.shape[0] != P.shape[0]:
        raise ValueError('nStepsOnOffList needs one (on, off) row per pressure field')
    sched = field_schedule(oo, int(TotalDurationSteps))
    return _bhte_run(P, sched, MaterialMap, MaterialList, dx, LocationMonitoring, nFactorMonitoring, dt, blood_rho, blood_ct,
                     stableTemp, 1.0, MonitoringPointsMap, initT0, initDose)

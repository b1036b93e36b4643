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

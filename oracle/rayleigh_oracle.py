"""CPU ORACLE (test infrastructure) for the Rayleigh-Sommerfeld sum that the reference obtains from
`BabelViscoFDTD.tools.RayleighAndBHTE.ForwardSimple` (package absent from /root/reference; call site
TranscranialModeling/BabelIntegrationSingle.py:295). PARITY UNPINNED against that package: the formula
below is the published Rayleigh integral for a baffled velocity source,
    u2(r) = (i k / 2 pi) sum_m u0_m dS_m exp(-i k R_m) / R_m ,
evaluated in float64 with numpy. The FDTD-vs-Rayleigh known-answer test (tests/test_oracle_physics.py,
K5) shows it reproduces the reference's own acceptance study together with the FDTD oracle.
Only tests/ may import this."""
import numpy as np


def ForwardSimple(cwvnb, center, ds, u0, rf, chunk=2048):
    k = complex(np.asarray(cwvnb).reshape(-1)[0])
    cen = np.asarray(center, np.float64).reshape(-1, 3)
    w = np.asarray(u0).reshape(-1).astype(np.complex128) * np.asarray(ds, np.float64).reshape(-1)
    pts = np.asarray(rf, np.float64).reshape(-1, 3)
    out = np.zeros(len(pts), np.complex128)
    for a in range(0, len(pts), chunk):
        R = np.sqrt(((pts[a:a + chunk, None, :] - cen[None, :, :]) ** 2).sum(axis=2))
        out[a:a + chunk] = (np.exp(-1j * k * R) / R) @ w
    return 1j * k / (2 * np.pi) * out


def ForwardSimpleC(cwvnb, center, ds, u0, rf, nthreads=0):
    """The same sum through oracle/rayleigh_oracle.c (OpenMP; float32 sine / cosine of the float64-reduced phase): for the
    volumes of the reference's study cases, which the numpy form above would take minutes for. Returns complex64."""
    import ctypes as C
    import os
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'librayleigh_oracle.so'))
    lib.bro_forward.argtypes = [C.c_long, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_long, C.c_void_p, C.c_void_p, C.c_int]
    k = complex(np.asarray(cwvnb).reshape(-1)[0])
    cen = np.ascontiguousarray(center, np.float32).reshape(-1, 3)
    dsf = np.ascontiguousarray(ds, np.float32).reshape(-1)
    u = np.ascontiguousarray(np.asarray(u0).reshape(-1).astype(np.complex64))
    pts = np.ascontiguousarray(rf, np.float32).reshape(-1, 3)
    out = np.zeros(len(pts), np.complex64)
    rc = lib.bro_forward(len(dsf), cen.ctypes.data, dsf.ctypes.data, u.view(np.float32).ctypes.data, k.real, k.imag, len(pts),
                         pts.ctypes.data, out.view(np.float32).ctypes.data, int(nthreads))
    if rc:
        raise MemoryError('bro_forward')
    return out

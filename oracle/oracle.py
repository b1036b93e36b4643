"""ctypes front-end of the CPU ORACLE (oracle/fdtd_oracle.c) -- test infrastructure only.

PARITY UNPINNED (see fdtd_oracle.c header): the reference's solver package
(BabelViscoFDTD==1.2.4) is absent from /root/reference, so this oracle restates its published
scheme and is validated by physics known-answer tests, not by reference outputs.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The signature mirrors the reference call sites (BabelIntegrationBASE.py:2338-2365) so parity
tests can feed the oracle and the HIP engine the very same arguments.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

MAP_BITS = {'Vx': 0, 'Vy': 1, 'Vz': 2, 'Sigmaxx': 3, 'Sigmayy': 4, 'Sigmazz': 5,
            'Sigmaxy': 6, 'Sigmaxz': 7, 'Sigmayz': 8, 'Pressure': 9, 'ALLV': 10}


class _Params(C.Structure):
    _fields_ = [('N1', C.c_int32), ('N2', C.c_int32), ('N3', C.c_int32), ('nMat', C.c_int32),
                ('NDelta', C.c_int32), ('nt', C.c_int32), ('typeSource', C.c_int32),
                ('lengthSource', C.c_int32), ('nSources', C.c_int32), ('sensorSub', C.c_int32),
                ('sensorStart', C.c_int32), ('selRMSorPeak', C.c_int32), ('selMapsRMS', C.c_uint32),
                ('selMapsSensors', C.c_uint32), ('qfactorCorrection', C.c_int32), ('nthreads', C.c_int32),
                ('h', C.c_double), ('dt', C.c_double), ('freq', C.c_double), ('reflectionLimit', C.c_double)]


def build(force=False):
    so = os.path.join(_HERE, 'libfdtd_oracle.so')
    src = os.path.join(_HERE, 'fdtd_oracle.c')
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(['make', '-C', _HERE, '-s'])
    return so


def default_threads():
    """Threads of a run that does not name a count: BABEL_ORACLE_THREADS, else at most 16 of the CPUs this process
    may use. The GPU boxes show 256 hardware threads; a static OpenMP team of that size on a 50^3 test grid spends
    its time in barriers (the bench's probe finds 16 fastest even at 384x384x256) and stalls for minutes when the
    box throttles the CPU share."""
    ev = os.environ.get('BABEL_ORACLE_THREADS')
    if ev:
        return max(int(ev), 1)
    return max(min(usable_cpus(), 16), 1)


def usable_cpus():
    """CPUs this process can really run on: affinity mask, capped by the cgroup CPU quota (cpu.max) if one is set."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(int(int(quota) / int(period)), 1))
    except (OSError, ValueError):
        pass
    return n


def lib():
    global _LIB
    if _LIB is None:
        # idle team members sleep instead of spinning (read by libgomp when it starts)
        os.environ.setdefault('OMP_WAIT_POLICY', 'PASSIVE')
        _LIB = C.CDLL(build())
        _LIB.bfo_stable_dt.restype = C.c_double
        _LIB.bfo_stable_dt.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_double, C.c_double]
        _LIB.bfo_run.restype = C.c_int
        _LIB.bfo_count_sensor_steps.restype = C.c_int
    return _LIB


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _xfast(a, dtype):
    """(N1,N2,N3) numpy array -> contiguous x-fastest buffer (shape (N3,N2,N1) C-order)."""
    return np.ascontiguousarray(np.asarray(a).transpose(2, 1, 0), dtype=dtype)


def _mask(names):
    m = 0
    for n in names:
        m |= 1 << MAP_BITS[n]
    return m


def _ordered(names):
    return sorted(set(names), key=lambda n: MAP_BITS[n])


def n_steps(TimeSimulation, dt):
    return int(np.ceil(TimeSimulation / dt - 1e-6))


def stable_dt(MaterialList, Frequency, QfactorCorrection, SpatialStep, AlphaCFL, QCorrection=1.0):
    ml = np.ascontiguousarray(MaterialList, np.float64)
    qc = np.ascontiguousarray(np.broadcast_to(np.asarray(QCorrection, np.float64), (ml.shape[0],)))
    return lib().bfo_stable_dt(ml.shape[0], _ptr(ml), _ptr(qc), float(Frequency), int(bool(QfactorCorrection)),
                               float(SpatialStep), float(AlphaCFL))


def tables(MaterialList, Frequency, SpatialStep, dt, QfactorCorrection=True, QCorrection=1.0):
    ml = np.ascontiguousarray(MaterialList, np.float64)
    qc = np.ascontiguousarray(np.broadcast_to(np.asarray(QCorrection, np.float64), (ml.shape[0],)))
    p = _Params(nMat=ml.shape[0], h=SpatialStep, dt=dt, freq=Frequency, qfactorCorrection=int(bool(QfactorCorrection)))
    out = np.zeros((7, ml.shape[0]), np.float32)
    c1k2 = np.zeros(2, np.float32)
    cmax = C.c_double()
    rc = lib().bfo_tables_f32(C.byref(p), _ptr(ml), _ptr(qc), _ptr(out), _ptr(c1k2), C.byref(cmax))
    assert rc == 0
    return out, c1k2, cmax.value


def cpml_profiles(N, NDelta, cmax, h, dt, freq, R):
    arrs = [np.zeros(N, np.float32) for _ in range(4)]
    lib().bfo_cpml_profiles(C.c_int(N), C.c_int(NDelta), C.c_double(cmax), C.c_double(h), C.c_double(dt),
                            C.c_double(freq), C.c_double(R), *[_ptr(a) for a in arrs])
    return arrs


def StaggeredFDTD_3D_with_relaxation(MaterialMap, MaterialList, Frequency, SourceMap, PulseSource,
                                     SpatialStep, TimeSimulation, SensorMap,
                                     Ox=np.array([1]), Oy=np.array([1]), Oz=np.array([1]),
                                     NDelta=12, DT=None, ReflectionLimit=1e-5, USE_SINGLE=True,
                                     SelMapsRMSPeakList=('Pressure',), SelMapsSensorsList=('Pressure',),
                                     SelRMSorPeak=1, AlphaCFL=1.0, TypeSource=0,
                                     QfactorCorrection=True, QCorrection=1.0,
                                     SensorSubSampling=2, SensorStart=0, ReflectorMask=None,
                                     nthreads=0, return_timing=False, **_ignored):
    """Same positional order and keyword meaning as BASE:2338-2365. Returns
    (Sensor, LastMap, DictRMS[, DictPeak], InputParam) like the reference's solver."""
    N1, N2, N3 = MaterialMap.shape
    N = N1 * N2 * N3
    ml = np.ascontiguousarray(MaterialList, np.float64)
    qc = np.ascontiguousarray(np.broadcast_to(np.asarray(QCorrection, np.float64), (ml.shape[0],)))
    if DT is None:
        DT = stable_dt(ml, Frequency, QfactorCorrection, SpatialStep, AlphaCFL, qc)
    nt = n_steps(TimeSimulation, DT)
    pulse = np.ascontiguousarray(np.atleast_2d(PulseSource), np.float64)
    selR = _ordered(SelMapsRMSPeakList)
    selS = _ordered(SelMapsSensorsList)
    p = _Params(N1=N1, N2=N2, N3=N3, nMat=ml.shape[0], NDelta=NDelta, nt=nt, typeSource=TypeSource,
                lengthSource=pulse.shape[1], nSources=pulse.shape[0], sensorSub=SensorSubSampling,
                sensorStart=SensorStart, selRMSorPeak=SelRMSorPeak, selMapsRMS=_mask(selR),
                selMapsSensors=_mask(selS), qfactorCorrection=int(bool(QfactorCorrection)), nthreads=nthreads or default_threads(),
                h=SpatialStep, dt=DT, freq=Frequency, reflectionLimit=ReflectionLimit)
    mm = _xfast(MaterialMap, np.uint32)
    sm = _xfast(SourceMap, np.uint32)
    sen = _xfast(SensorMap, np.uint32)
    refl = None if ReflectorMask is None else _xfast(ReflectorMask, np.uint32)

    def weight(o):
        o = np.asarray(o)
        if o.size == 1:
            return None if float(o.reshape(-1)[0]) == 1.0 else np.full((N3, N2, N1), float(o.reshape(-1)[0]), np.float64)
        return _xfast(o, np.float64)
    ox, oy, oz = weight(Ox), weight(Oy), weight(Oz)
    nSens = int(np.count_nonzero(sen))
    nTs = lib().bfo_count_sensor_steps(nt, SensorSubSampling, SensorStart)
    sensors = np.zeros((len(selS), nSens, max(nTs, 0)), np.float32)
    idx = np.zeros(nSens, np.uint32)
    rms = np.zeros((len(selR), N), np.float32) if SelRMSorPeak & 1 else None
    peak = np.zeros((len(selR), N), np.float32) if SelRMSorPeak & 2 else None
    last = np.zeros((len(selR), N), np.float32)
    secs = C.c_double()
    rc = lib().bfo_run(C.byref(p), _ptr(mm), _ptr(ml), _ptr(qc), _ptr(sm), _ptr(pulse), _ptr(ox), _ptr(oy), _ptr(oz),
                       _ptr(sen), _ptr(refl), _ptr(sensors), _ptr(idx), _ptr(rms), _ptr(peak), _ptr(last),
                       C.byref(secs))
    if rc != 0:
        raise RuntimeError('oracle bfo_run failed rc=%d' % rc)

    def vol(a):
        return np.ascontiguousarray(a.reshape(N3, N2, N1).transpose(2, 1, 0))
    steps = np.arange(nt)
    keep = steps[(steps % SensorSubSampling == 0) & (steps // SensorSubSampling >= SensorStart)]
    Sensor = {'time': keep * DT}
    for q, name in enumerate(selS):
        Sensor[name] = sensors[q]
    LastMap = {name: vol(last[q]) for q, name in enumerate(selR)}
    InputParam = {'IndexSensorMap': idx, 'DT': DT, 'nt': nt, 'stepLoopSeconds': secs.value}
    out = [Sensor, LastMap]
    if SelRMSorPeak & 1:
        out.append({name: vol(rms[q]) for q, name in enumerate(selR)})
    if SelRMSorPeak & 2:
        out.append({name: vol(peak[q]) for q, name in enumerate(selR)})
    out.append(InputParam)
    return tuple(out)


# ------------------------------------------------------------------------------------------------
# Z-slab form of the oracle: same interface as babelbrain_amd.slab.HipSlab, so the slab
# decomposition and its halo exchange (babelbrain_amd/slab.py) can be exercised on CPU with the
# gloo backend (tests/test_slab_gloo.py).
# ------------------------------------------------------------------------------------------------
class OracleSlab:
    def __init__(self, args, kwargs, k0, nk, nthreads=0):
        import torch
        self.torch = torch
        MaterialMap, MaterialList, Frequency, SourceMap, PulseSource, SpatialStep, Duration, SensorMap = args
        N1, N2, N3 = MaterialMap.shape
        self.N1, self.N2, self.N3, self.k0, self.nk = N1, N2, N3, k0, nk
        ml = np.ascontiguousarray(MaterialList, np.float64)
        qc = np.ascontiguousarray(np.broadcast_to(np.asarray(kwargs.get('QCorrection', 1.0), np.float64), (ml.shape[0],)))
        DT = kwargs['DT']
        self.DT = DT
        self.nt = n_steps(Duration, DT)
        pulse = np.ascontiguousarray(np.atleast_2d(PulseSource), np.float64)
        self.selR = _ordered(kwargs.get('SelMapsRMSPeakList', ('Pressure',)))
        self.selS = _ordered(kwargs.get('SelMapsSensorsList', ('Pressure',)))
        self.mode = kwargs.get('SelRMSorPeak', 1)
        self.sub, self.start = kwargs.get('SensorSubSampling', 1), kwargs.get('SensorStart', 0)
        p = _Params(N1=N1, N2=N2, N3=N3, nMat=ml.shape[0], NDelta=kwargs.get('NDelta', 12), nt=self.nt,
                    typeSource=kwargs.get('TypeSource', 0), lengthSource=pulse.shape[1], nSources=pulse.shape[0],
                    sensorSub=self.sub, sensorStart=self.start, selRMSorPeak=self.mode, selMapsRMS=_mask(self.selR),
                    selMapsSensors=_mask(self.selS), qfactorCorrection=int(bool(kwargs.get('QfactorCorrection', True))),
                    nthreads=nthreads or default_threads(), h=SpatialStep, dt=DT, freq=Frequency,
                    reflectionLimit=kwargs.get('ReflectionLimit', 1e-5))
        gl, gh = min(2, k0), min(2, N3 - (k0 + nk))
        mm = _xfast(np.asarray(MaterialMap)[:, :, k0 - gl:k0 + nk + gh], np.uint32)
        sl = slice(k0, k0 + nk)
        sm = _xfast(np.asarray(SourceMap)[:, :, sl], np.uint32)
        sen = _xfast(np.asarray(SensorMap)[:, :, sl], np.uint32)
        refl = kwargs.get('ReflectorMask')
        refl = None if refl is None else _xfast(np.asarray(refl)[:, :, sl], np.uint32)

        def weight(o):
            o = np.asarray(o)
            if o.size == 1:
                v = float(o.reshape(-1)[0])
                return None if v == 1.0 else np.full((nk, N2, N1), v, np.float64)
            return _xfast(o[:, :, sl], np.float64)
        one = np.array([1])
        ox, oy, oz = (weight(kwargs.get(n, one)) for n in ('Ox', 'Oy', 'Oz'))
        L = lib()
        L.bfo_create.restype = C.c_void_p
        rc = C.c_int()
        self.h = L.bfo_create(C.byref(p), C.c_int(k0), C.c_int(nk), _ptr(mm), C.c_int(gl), C.c_int(gh), _ptr(ml), _ptr(qc),
                              _ptr(sm), _ptr(pulse), _ptr(ox), _ptr(oy), _ptr(oz), _ptr(sen), _ptr(refl), C.byref(rc))
        if not self.h:
            raise RuntimeError('bfo_create failed rc=%d' % rc.value)
        self.h = C.c_void_p(self.h)
        self._buf = {}
        for g in (0, 1):
            for f in range(3):
                for side in (0, 1):
                    for send in (0, 1):
                        self._buf[(g, f, side, send)] = torch.zeros(2 * N1 * N2, dtype=torch.float32)

    def halo(self, group, f, side, send):
        return self._buf[(group, f, side, int(send))]

    def before_send(self, group):
        for f in range(3):
            for side in (0, 1):
                b = self._buf[(group, f, side, 1)]
                lib().bfo_halo_get(self.h, group, f, side, C.c_void_p(b.data_ptr()))

    def after_recv(self, group, sides):
        for f in range(3):
            for side in sides:
                b = self._buf[(group, f, side, 0)]
                lib().bfo_halo_put(self.h, group, f, side, C.c_void_p(b.data_ptr()))

    def half_step_stress(self):
        lib().bfo_half_stress(self.h)

    def half_step_velocity(self):
        lib().bfo_half_velocity(self.h)

    def sync(self):
        pass

    def outputs(self):
        L = lib()
        L.bfo_num_sensors.restype = C.c_size_t
        nS = L.bfo_num_sensors(self.h)
        nTs = L.bfo_num_sensor_steps(self.h)
        N = self.N1 * self.N2 * self.nk
        sens = np.zeros((len(self.selS), nS, nTs), np.float32)
        idx = np.zeros(nS, np.uint32)
        rms = np.zeros((len(self.selR), N), np.float32) if self.mode & 1 else None
        peak = np.zeros((len(self.selR), N), np.float32) if self.mode & 2 else None
        last = np.zeros((len(self.selR), N), np.float32)
        L.bfo_results(self.h, _ptr(sens), _ptr(idx), _ptr(rms), _ptr(peak), _ptr(last))

        def vol(a):
            return np.ascontiguousarray(a.reshape(self.nk, self.N2, self.N1).transpose(2, 1, 0))
        steps = np.arange(self.nt)
        keep = steps[(steps % self.sub == 0) & (steps // self.sub >= self.start)]
        out = {'Sensor': {'time': keep * self.DT}, 'IndexSensorMap': idx,
               'LastMap': {n: vol(last[q]) for q, n in enumerate(self.selR)}}
        for q, n in enumerate(self.selS):
            out['Sensor'][n] = sens[q]
        if self.mode & 1:
            out['RMS'] = {n: vol(rms[q]) for q, n in enumerate(self.selR)}
        if self.mode & 2:
            out['Peak'] = {n: vol(peak[q]) for q, n in enumerate(self.selR)}
        return out

    def close(self):
        if self.h:
            lib().bfo_destroy(self.h)
            self.h = None

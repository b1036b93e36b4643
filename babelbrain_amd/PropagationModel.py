"""Drop-in for `BabelViscoFDTD.PropagationModel.PropagationModel` on AMD MI355X.

The reference instantiates the solver once as a module global and calls two methods on it:

    PModel = PropagationModel()                                       BabelIntegrationBASE.py:43
    PModel.CalculateMatricesForPropagation(map, MatArray, f, Qcorr, dx, AlphaCFL)   BASE:1799,1801
    PModel.StaggeredFDTD_3D_with_relaxation(MaterialMap, MaterialList, Frequency, SourceMap,
                                            PulseSource, SpatialStep, TimeSimulation, SensorMap, **kw)
                                                                      BASE:2338,2374,2401

This class keeps those names, the positional order, the keyword names and the return shapes, and
runs the time loop on the hand-written HIP engine (libbabelfdtd_hip.so) through ctypes.
Errors surface as Python exceptions, which CalculateFieldProcess.py:126-129 turns into its
`--Babel-Brain-Low-Error` sentinel. There is no CPU fallback.
"""
import numpy as np

from . import _engine
from ._engine import Engine, EngineError, KIND_LAST, KIND_PEAK, KIND_RMS

COMPUTING_BACKEND_HIP = 5   # new code next to 0 CPU, 1 CUDA, 2 OpenCL, 3 Metal, 4 MLX (BabelBrain.py:429-439)


def n_steps(TimeSimulation, dt):
    """Number of time steps of a run of duration TimeSimulation (the caller builds it as nt*dt, BASE:2089)."""
    return int(np.ceil(TimeSimulation / dt - 1e-6))


def sensor_steps(nt, SensorSubSampling, SensorStart):
    steps = np.arange(nt)
    return steps[(steps % SensorSubSampling == 0) & (steps // SensorSubSampling >= SensorStart)]


def compact_sources(SourceMap, Ox, Oy, Oz, k0=0, nk=None):
    """SourceMap (uint32, 0 = none, s>=1 -> PulseSource row s-1; Single:326-344) and the per-voxel
    weights Ox,Oy,Oz (full volume, or size-1 arrays; BASE:2325-2335) -> compact lists for one Z-slab."""
    N1, N2, N3 = SourceMap.shape
    nk = N3 - k0 if nk is None else nk
    sub = SourceMap[:, :, k0:k0 + nk]
    # sources sit on one or a few z planes (Single:326-344): look for the planes first, then for the voxels inside them
    # (a 3-D nonzero over the whole volume costs 0.3 s at 512^3)
    planes = np.flatnonzero(sub.any(axis=(0, 1)))
    if planes.size and planes.size * 8 <= nk:
        parts = [np.nonzero(sub[:, :, kp]) for kp in planes]
        ii = np.concatenate([p[0] for p in parts]); jj = np.concatenate([p[1] for p in parts])
        kk = np.concatenate([np.full(p[0].shape, kp, np.int64) for p, kp in zip(parts, planes)])
    else:
        ii, jj, kk = np.nonzero(sub)
    lin = (ii.astype(np.int64) + N1 * (jj.astype(np.int64) + N2 * kk.astype(np.int64)))
    order = np.argsort(lin, kind='stable')
    ii, jj, kk, lin = ii[order], jj[order], kk[order], lin[order]
    row = sub[ii, jj, kk].astype(np.int64) - 1

    def w(o):
        o = np.asarray(o)
        if o.size == 1:
            v = float(o.reshape(-1)[0])
            return None if v == 1.0 else np.full(lin.shape, v, np.float32)
        if o.shape != SourceMap.shape:
            raise ValueError('Ox/Oy/Oz must be size-1 or have the shape of the domain')
        return o[ii, jj, kk + k0].astype(np.float32)
    # int64: a multi-device domain may hold more than 2^32 voxels; the single-engine ABI takes uint32 (Engine.set_sources checks)
    return lin.astype(np.int64), row.astype(np.uint32), w(Ox), w(Oy), w(Oz)


def material_slab(MaterialMap, k0, nk):
    """Slab view plus up to 2 ghost planes each side, as bfd_set_material_map wants it."""
    N3 = MaterialMap.shape[2]
    gl = min(2, k0)
    gh = min(2, N3 - (k0 + nk))
    return MaterialMap[:, :, k0 - gl:k0 + nk + gh], gl, gh


def _device_list(spec):
    """'all' | '0,1,2' | iterable of ordinals -> list of HIP device ordinals (an ordinal may repeat: slabs sharing a device)."""
    if spec is None:
        return None
    if isinstance(spec, str):
        spec = spec.strip()
        if not spec:
            return None
        if spec.lower() == 'all':
            return [d for d, _ in _engine.list_devices()]
        return [int(x) for x in spec.split(',')]
    if np.isscalar(spec):
        return [int(spec)]
    return [int(x) for x in spec]


class PropagationModel:
    def __init__(self, device=None, kernelVariant=0, devices=None, keepPlacementCache=None):
        """device: HIP ordinal of a single-device run. devices: list of ordinals (or 'all'): ONE call is then split into
        Z-slabs over those devices inside the library (bfd_group_*), the return values are the whole-domain ones. The
        environment variable BABELFDTD_DEVICES ('all' or '0,1,2,3') does the same for a caller that cannot pass arguments
        (the reference builds its module-global PModel without any, BASE:43).
        keepPlacementCache: the library keeps the device buffers a placement search found (bfd_placement_cache_*) for the next
        engine of the process. By default a solver call gives them back when it returns, so that nothing idle of this library
        sits on the device beside whatever the process does next (a thermal step, another library); True (or
        BABELFDTD_PLACEMENT_CACHE_KEEP=1) keeps them between calls -- what the two or three back-to-back calls of one
        RUN_SIMULATION (BASE:2338, 2374, 2401) want; release them with _engine.placement_cache_release()."""
        import os
        if keepPlacementCache is None:
            keepPlacementCache = os.environ.get('BABELFDTD_PLACEMENT_CACHE_KEEP', '0') not in ('', '0')
        self._keepPlacementCache = bool(keepPlacementCache)
        self._device = device
        self._devices = _device_list(devices if devices is not None else os.environ.get('BABELFDTD_DEVICES'))
        self._kernelVariant = kernelVariant
        self.last_timing = None

    # ------------------------------------------------------------------------------------------
    def CalculateMatricesForPropagation(self, MaterialMap, MaterialProperties, Frequency, QfactorCorrection, h,
                                        AlphaCFL, QCorrection=1.0):
        """Returns a 10-tuple whose element 0 is the stable time step, like the reference's solver
        (only element 0 is read: BASE:1799, 1801). Elements 1-7 are this engine's float32
        per-material tables for that time step, 8 = (c1,k2), 9 = fastest wave speed."""
        dt = _engine.stable_dt(MaterialProperties, Frequency, QfactorCorrection, h, AlphaCFL, QCorrection)
        t, c1k2, cmax = _engine.material_tables(MaterialProperties, Frequency, QfactorCorrection, h, dt, QCorrection)
        return (dt, t[0], t[1], t[2], t[3], t[4], t[5], t[6], c1k2, cmax)

    # ------------------------------------------------------------------------------------------
    def StaggeredFDTD_3D_with_relaxation(self, MaterialMap, MaterialProperties, Frequency, SourceMap, PulseSource,
                                         SpatialStep, DurationSimulation, SensorMap,
                                         Ox=np.array([1]), Oy=np.array([1]), Oz=np.array([1]),
                                         AlphaCFL=0.99, NDelta=12, ReflectionLimit=1.0000e-05,
                                         COMPUTING_BACKEND=COMPUTING_BACKEND_HIP, USE_SINGLE=True, DT=None,
                                         QfactorCorrection=True, QCorrection=1.0, TypeSource=0, SelRMSorPeak=1,
                                         SelMapsRMSPeakList=('ALLV',), SelMapsSensorsList=('Vx', 'Vy', 'Vz'),
                                         SensorSubSampling=2, SensorStart=0, DefaultGPUDeviceName='MI355X',
                                         DefaultGPUDeviceNumber=None, ReflectorMask=None, SILENT=False,
                                         ReturnSensorDFT=False, ReturnSensorSeries=True, **unused):
        """Extensions beyond the reference's keyword set: ReturnSensorDFT=True adds the single-frequency
        content of the sensor series (what CalculatePhaseData extracts, BASE:2498-2520) to InputParam; with
        ReturnSensorSeries=False as well, the series themselves are never stored -- re/im/peak are accumulated per sensor
        while the samples are taken (20 B per sensor instead of 4*nTs) and Sensor holds only 'time'."""
        if not USE_SINGLE:
            raise NotImplementedError('the MI355X engine computes in float32 (USE_SINGLE=True, BASE:2354)')
        MaterialMap = np.asarray(MaterialMap)
        if MaterialMap.ndim != 3:
            raise ValueError('MaterialMap must be a 3-D array')
        for nm, a in (('SourceMap', SourceMap), ('SensorMap', SensorMap)):
            if np.asarray(a).shape != MaterialMap.shape:
                raise ValueError('%s must have the shape of MaterialMap' % nm)
        N1, N2, N3 = MaterialMap.shape
        ml = np.ascontiguousarray(MaterialProperties, np.float64).reshape(-1, 5)
        if DT is None:
            DT = _engine.stable_dt(ml, Frequency, QfactorCorrection, SpatialStep, AlphaCFL, QCorrection)
        nt = n_steps(DurationSimulation, DT)
        # device: explicit ordinal of the call, else the one this object was built with, else the first device whose
        # name contains DefaultGPUDeviceName (the reference selects by name substring, BASE:2358, 918-925), else 0
        devices = self._devices
        if DefaultGPUDeviceNumber is not None and not np.isscalar(DefaultGPUDeviceNumber):
            devices, DefaultGPUDeviceNumber = _device_list(DefaultGPUDeviceNumber), None
        if DefaultGPUDeviceNumber is None and self._device is None and devices is not None and len(devices) >= 1:
            devices = devices[:max(1, N3 // 4)]            # a slab owns at least the 2 + 2 planes its neighbours read
            if len(devices) > 1:
                return self._run_group(devices, MaterialMap, ml, Frequency, SourceMap, PulseSource, SpatialStep, DT, nt, SensorMap,
                                       Ox, Oy, Oz, NDelta, ReflectionLimit, QfactorCorrection, QCorrection, TypeSource, SelRMSorPeak,
                                       SelMapsRMSPeakList, SelMapsSensorsList, SensorSubSampling, SensorStart, ReflectorMask, SILENT,
                                       ReturnSensorDFT, ReturnSensorSeries)
            DefaultGPUDeviceNumber = devices[0]
        device = DefaultGPUDeviceNumber if DefaultGPUDeviceNumber is not None else self._device
        if device is None:
            device = 0
            for d, name in _engine.list_devices() if DefaultGPUDeviceName else []:
                if str(DefaultGPUDeviceName).lower() in name.lower():
                    device = d
                    break
        eng = Engine(N1, N2, N3, ml.shape[0], SpatialStep, DT, Frequency, nt, NDelta=NDelta,
                     reflectionLimit=ReflectionLimit, typeSource=TypeSource, sensorSub=SensorSubSampling,
                     sensorStart=SensorStart, selRMSorPeak=SelRMSorPeak, selMapsRMS=SelMapsRMSPeakList,
                     selMapsSensors=SelMapsSensorsList, qfactorCorrection=QfactorCorrection, device=device,
                     kernelVariant=self._kernelVariant, sensorMode=1 if (ReturnSensorDFT and not ReturnSensorSeries) else 0)
        if not ReturnSensorSeries and not ReturnSensorDFT:
            eng.close()
            raise ValueError('ReturnSensorSeries=False needs ReturnSensorDFT=True')
        try:
            eng.set_materials(ml, QCorrection)
            eng.set_material_map(MaterialMap, 0, 0)
            if ReflectorMask is not None:
                eng.set_reflector(ReflectorMask)
            lin, row, wx, wy, wz = compact_sources(np.asarray(SourceMap), Ox, Oy, Oz)
            eng.set_sources(lin, row, wx, wy, wz, PulseSource)
            eng.set_sensor_map(SensorMap)
            eng.timing_begin(False)
            eng.run(nt)
            self.last_timing = eng.timing_end()
            self.last_timing['voxel_steps'] = float(N1) * N2 * N3 * nt
            Sensor = {'time': sensor_steps(nt, SensorSubSampling, SensorStart) * DT}
            if ReturnSensorSeries:
                sens = eng.sensors()
                for q, name in enumerate(eng.selS):
                    Sensor[name] = sens[q]
            InputParam = {'IndexSensorMap': eng.sensor_index(), 'DT': DT, 'nt': nt,
                          'device_bytes': eng.device_bytes, 'timing': self.last_timing, 'placement': eng.placement_note()}
            if ReturnSensorDFT:
                # extension: what CalculatePhaseData extracts from the sensor block
                # (BASE:2498-2520), computed on the device
                F, pk = eng.sensor_dft(Frequency)
                InputParam['SensorDFT'] = {name: F[q] for q, name in enumerate(eng.selS)}
                InputParam['SensorPeak'] = {name: pk[q] for q, name in enumerate(eng.selS)}
            LastMap = {name: eng.get_map(KIND_LAST, name) for name in eng.selR}
            out = [Sensor, LastMap]
            if SelRMSorPeak & 1:
                out.append({name: eng.get_map(KIND_RMS, name) for name in eng.selR})
            if SelRMSorPeak & 2:
                out.append({name: eng.get_map(KIND_PEAK, name) for name in eng.selR})
            out.append(InputParam)
        finally:
            eng.close()
            if not self._keepPlacementCache:
                _engine.placement_cache_release()
        if not SILENT and self.last_timing['total_ms'] > 0:
            print('HIP FDTD: %d steps, %.1f Mvoxel-steps/s' % (
                nt, self.last_timing['voxel_steps'] / self.last_timing['total_ms'] / 1e3))
        return tuple(out)

    # ------------------------------------------------------------------------------------------
    def _run_group(self, devices, MaterialMap, ml, Frequency, SourceMap, PulseSource, SpatialStep, DT, nt, SensorMap, Ox, Oy, Oz,
                   NDelta, ReflectionLimit, QfactorCorrection, QCorrection, TypeSource, SelRMSorPeak, SelMapsRMSPeakList,
                   SelMapsSensorsList, SensorSubSampling, SensorStart, ReflectorMask, SILENT, ReturnSensorDFT, ReturnSensorSeries):
        """The same call on several devices: whole-domain arrays in, whole-domain results out (bfd_group_*)."""
        if not ReturnSensorSeries and not ReturnSensorDFT:
            raise ValueError('ReturnSensorSeries=False needs ReturnSensorDFT=True')
        N1, N2, N3 = MaterialMap.shape
        grp = _engine.Group(devices, N1, N2, N3, ml.shape[0], SpatialStep, DT, Frequency, nt, NDelta=NDelta,
                            reflectionLimit=ReflectionLimit, typeSource=TypeSource, sensorSub=SensorSubSampling,
                            sensorStart=SensorStart, selRMSorPeak=SelRMSorPeak, selMapsRMS=SelMapsRMSPeakList,
                            selMapsSensors=SelMapsSensorsList, qfactorCorrection=QfactorCorrection,
                            kernelVariant=self._kernelVariant, sensorMode=1 if (ReturnSensorDFT and not ReturnSensorSeries) else 0)
        try:
            grp.set_materials(ml, QCorrection)
            grp.set_material_map(MaterialMap)
            if ReflectorMask is not None:
                grp.set_reflector(ReflectorMask)
            lin, row, wx, wy, wz = compact_sources(np.asarray(SourceMap), Ox, Oy, Oz)
            grp.set_sources(lin, row, wx, wy, wz, PulseSource)
            grp.set_sensor_map(SensorMap)
            grp.timing_begin()
            grp.run(nt)
            self.last_timing = grp.timing_end()
            self.last_timing['voxel_steps'] = float(N1) * N2 * N3 * nt
            Sensor = {'time': sensor_steps(nt, SensorSubSampling, SensorStart) * DT}
            if ReturnSensorSeries:
                sens = grp.sensors()
                for q, name in enumerate(grp.selS):
                    Sensor[name] = sens[q]
            InputParam = {'IndexSensorMap': grp.sensor_index(), 'DT': DT, 'nt': nt, 'device_bytes': grp.device_bytes,
                          'timing': self.last_timing, 'devices': list(devices),
                          'slabs': [grp.slab(r)[:3] for r in range(grp.size)],
                          'placement': [grp.slab(r)[3].placement_note() for r in range(grp.size)]}
            if ReturnSensorDFT:
                F, pk = grp.sensor_dft(Frequency)
                InputParam['SensorDFT'] = {name: F[q] for q, name in enumerate(grp.selS)}
                InputParam['SensorPeak'] = {name: pk[q] for q, name in enumerate(grp.selS)}
            LastMap = {name: grp.get_map(KIND_LAST, name) for name in grp.selR}
            out = [Sensor, LastMap]
            if SelRMSorPeak & 1:
                out.append({name: grp.get_map(KIND_RMS, name) for name in grp.selR})
            if SelRMSorPeak & 2:
                out.append({name: grp.get_map(KIND_PEAK, name) for name in grp.selR})
            out.append(InputParam)
        finally:
            grp.close()
            if not self._keepPlacementCache:
                _engine.placement_cache_release()
        if not SILENT and self.last_timing['total_ms'] > 0:
            print('HIP FDTD: %d steps on %d slabs, %.1f Mvoxel-steps/s' % (
                nt, len(devices), self.last_timing['voxel_steps'] / self.last_timing['total_ms'] / 1e3))
        return tuple(out)

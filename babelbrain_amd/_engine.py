"""ctypes binding of libbabelfdtd_hip.so (include/babelfdtd.h).

This is the thin layer BASELINE.json's north_star asks for: the reference's Python driver
(TranscranialModeling/BabelIntegrationBASE.py:2338) reaches the HIP engine through it.
There is no CPU fallback: a missing library or a missing GPU raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# BABELFDTD_HIP_LIB: another build of the same library (A/B kernel experiments, scripts/ab_build.sh)
LIB_PATH = os.environ.get('BABELFDTD_HIP_LIB') or os.path.join(_HERE, 'libbabelfdtd_hip.so')

MAP_BITS = {'Vx': 0, 'Vy': 1, 'Vz': 2, 'Sigmaxx': 3, 'Sigmayy': 4, 'Sigmazz': 5,
            'Sigmaxy': 6, 'Sigmaxz': 7, 'Sigmayz': 8, 'Pressure': 9, 'ALLV': 10}
KIND_RMS, KIND_PEAK, KIND_LAST = 0, 1, 2
HALO_VELOCITY, HALO_STRESS = 0, 1
KERNEL_CLASSES = ['stress_fluid', 'stress_normal_solid', 'stress_shear_sparse', 'velocity_fluid', 'velocity_solid', 'fused_fluid']
FIELD_NAMES = ['Vx', 'Vy', 'Vz', 'Sxx', 'Syy', 'Szz', 'Sxy', 'Sxz', 'Syz', 'Rxx', 'Ryy', 'Rzz', 'Rxy', 'Rxz', 'Ryz']

# every symbol include/babelfdtd.h declares (tests check that the library exports all of them)
ABI_SYMBOLS = [
    'bfd_abi_version', 'bfd_last_error', 'bfd_device_count', 'bfd_device_name', 'bfd_stable_dt',
    'bfd_material_tables', 'bfd_create', 'bfd_destroy', 'bfd_set_stream', 'bfd_use_private_stream', 'bfd_set_materials',
    'bfd_set_material_map', 'bfd_set_reflector', 'bfd_set_sources', 'bfd_set_sensor_map', 'bfd_run',
    'bfd_half_step_stress', 'bfd_half_step_velocity', 'bfd_half_step_stress_part', 'bfd_half_step_velocity_part', 'bfd_half_step_stress_part_on', 'bfd_half_step_velocity_part_on', 'bfd_sync', 'bfd_current_step', 'bfd_prepare', 'bfd_halo_region',
    'bfd_timing_begin', 'bfd_timing_end', 'bfd_timing_kernels', 'bfd_algorithmic_bytes', 'bfd_reset', 'bfd_num_sensors', 'bfd_num_sensor_steps', 'bfd_get_sensor_index',
    'bfd_get_sensors', 'bfd_get_map', 'bfd_get_field', 'bfd_tile_counts', 'bfd_tile_count_lean', 'bfd_tile_count_fused', 'bfd_activity_counts', 'bfd_device_bytes', 'bfd_rayleigh_forward', 'bfd_get_sensor_dft', 'bfd_dft_series', 'bfd_bhte_run', 'bfd_bhte_run_fields', 'bfd_bhte_run_volumes',
    'bfd_halo_fields', 'bfd_placement_note', 'bfd_set_placement', 'bfd_group_set_placement',
    'bfd_group_create', 'bfd_group_destroy', 'bfd_group_size', 'bfd_group_slab', 'bfd_group_set_materials', 'bfd_group_set_material_map',
    'bfd_group_set_reflector', 'bfd_group_set_sources', 'bfd_group_set_sensor_map', 'bfd_group_prepare', 'bfd_group_run', 'bfd_group_sync',
    'bfd_group_reset', 'bfd_group_timing_begin', 'bfd_group_timing_end', 'bfd_group_num_sensors', 'bfd_group_num_sensor_steps',
    'bfd_group_get_sensor_index', 'bfd_group_get_sensors', 'bfd_group_get_sensor_dft', 'bfd_group_get_map', 'bfd_group_device_bytes',
    'bfd_group_peer_status', 'bfd_placement_cache_release',
]


class Config(C.Structure):
    _fields_ = [('N1', C.c_int32), ('N2', C.c_int32), ('N3', C.c_int32), ('k0', C.c_int32), ('nk', C.c_int32),
                ('nMat', C.c_int32), ('NDelta', C.c_int32), ('typeSource', C.c_int32), ('sensorSub', C.c_int32),
                ('sensorStart', C.c_int32), ('nt', C.c_int32), ('selRMSorPeak', C.c_int32),
                ('selMapsRMS', C.c_uint32), ('selMapsSensors', C.c_uint32), ('qfactorCorrection', C.c_int32),
                ('device', C.c_int32), ('kernelVariant', C.c_int32), ('rmsFirstStep', C.c_int32), ('sensorMode', C.c_int32),
                ('h', C.c_double), ('dt', C.c_double), ('freq', C.c_double), ('reflectionLimit', C.c_double)]


class EngineError(RuntimeError):
    pass


_lib = None


def _preload_torch_hip_runtime():
    """One HIP runtime per process. PyTorch-ROCm wheels bundle their own libamdhip64 (soname
    libamdhip64.so.7, the one this library needs too). If this library were loaded first it would bind to
    the system runtime and a later `import torch` would bring a second one, which then finds no GPU. So when
    a torch wheel is installed its runtime is loaded first (a plain dlopen, torch itself is not imported);
    multi-GPU runs (slab.py) hand torch streams and RCCL the engine's device memory and need the shared runtime."""
    import importlib.util
    import sys
    if 'torch' in sys.modules:
        return
    try:
        spec = importlib.util.find_spec('torch')
    except Exception:
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], 'lib', 'libamdhip64.so')
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load_library():
    """Load libbabelfdtd_hip.so; raises if it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineError('HIP engine library missing: %s (build it with `make -C babelbrain_amd/csrc`); '
                          'there is no CPU fallback' % LIB_PATH)
    _preload_torch_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    lib.bfd_last_error.restype = C.c_char_p
    lib.bfd_stable_dt.restype = C.c_double
    lib.bfd_stable_dt.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_double, C.c_int32, C.c_double, C.c_double]
    lib.bfd_material_tables.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_double, C.c_int32, C.c_double,
                                        C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.bfd_create.argtypes = [C.POINTER(Config), C.POINTER(C.c_void_p)]
    lib.bfd_destroy.argtypes = [C.c_void_p]
    lib.bfd_destroy.restype = None
    lib.bfd_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    lib.bfd_use_private_stream.argtypes = [C.c_void_p]
    lib.bfd_set_materials.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.bfd_set_material_map.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32]
    lib.bfd_set_reflector.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64]
    lib.bfd_set_sources.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_int32, C.c_int32]
    lib.bfd_set_sensor_map.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_int64)]
    lib.bfd_run.argtypes = [C.c_void_p, C.c_int32]
    lib.bfd_half_step_stress.argtypes = [C.c_void_p]
    lib.bfd_half_step_velocity.argtypes = [C.c_void_p]
    lib.bfd_half_step_stress_part.argtypes = [C.c_void_p, C.c_int32]
    lib.bfd_half_step_velocity_part.argtypes = [C.c_void_p, C.c_int32]
    lib.bfd_half_step_stress_part_on.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    lib.bfd_half_step_velocity_part_on.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    lib.bfd_sync.argtypes = [C.c_void_p]
    lib.bfd_current_step.argtypes = [C.c_void_p]
    lib.bfd_prepare.argtypes = [C.c_void_p]
    lib.bfd_halo_region.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p),
                                    C.POINTER(C.c_size_t)]
    lib.bfd_timing_begin.argtypes = [C.c_void_p, C.c_int32]
    lib.bfd_timing_end.argtypes = [C.c_void_p] + [C.POINTER(C.c_double)] * 4 + [C.POINTER(C.c_int64)] * 2
    lib.bfd_timing_kernels.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.bfd_algorithmic_bytes.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    lib.bfd_reset.argtypes = [C.c_void_p]
    lib.bfd_num_sensors.argtypes = [C.c_void_p]
    lib.bfd_num_sensors.restype = C.c_int64
    lib.bfd_num_sensor_steps.argtypes = [C.c_void_p]
    lib.bfd_get_sensor_index.argtypes = [C.c_void_p, C.c_void_p]
    lib.bfd_get_sensors.argtypes = [C.c_void_p, C.c_void_p]
    lib.bfd_get_map.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_int64, C.c_int64]
    lib.bfd_get_field.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int64, C.c_int64]
    lib.bfd_device_bytes.argtypes = [C.c_void_p]
    lib.bfd_tile_counts.argtypes = [C.c_void_p] + [C.POINTER(C.c_int32)] * 5
    lib.bfd_tile_count_lean.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    lib.bfd_tile_count_fused.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    lib.bfd_activity_counts.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    lib.bfd_device_bytes.restype = C.c_int64
    lib.bfd_device_name.argtypes = [C.c_int, C.c_char_p, C.c_int]
    lib.bfd_bhte_run.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p, C.c_float, C.c_double, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                 C.c_int64, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
    lib.bfd_bhte_run_fields.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_double, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                        C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
    lib.bfd_bhte_run_volumes.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_double, C.c_int32, C.c_void_p,
                                         C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
    lib.bfd_get_sensor_dft.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
    lib.bfd_dft_series.argtypes = [C.c_int32, C.c_int64, C.c_int32, C.c_void_p, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
    lib.bfd_rayleigh_forward.argtypes = [C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double,
                                         C.c_int64, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
    lib.bfd_halo_fields.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_uint32)]
    lib.bfd_placement_note.argtypes = [C.c_void_p]
    lib.bfd_placement_note.restype = C.c_char_p
    lib.bfd_set_placement.argtypes = [C.c_void_p, C.c_int32, C.c_int64]
    lib.bfd_group_set_placement.argtypes = [C.c_void_p, C.c_int32, C.c_int64]
    lib.bfd_group_create.argtypes = [C.POINTER(Config), C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]
    lib.bfd_group_destroy.argtypes = [C.c_void_p]
    lib.bfd_group_destroy.restype = None
    lib.bfd_group_size.argtypes = [C.c_void_p]
    lib.bfd_group_slab.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_void_p)]
    lib.bfd_group_set_materials.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.bfd_group_set_material_map.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64]
    lib.bfd_group_set_reflector.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64]
    lib.bfd_group_set_sources.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_int32, C.c_int32]
    lib.bfd_group_set_sensor_map.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_int64)]
    for fn in ('bfd_group_prepare', 'bfd_group_sync', 'bfd_group_reset', 'bfd_group_timing_begin'):
        getattr(lib, fn).argtypes = [C.c_void_p]
    lib.bfd_group_run.argtypes = [C.c_void_p, C.c_int32]
    lib.bfd_group_timing_end.argtypes = [C.c_void_p] + [C.POINTER(C.c_double)] * 4 + [C.POINTER(C.c_int32)]
    lib.bfd_group_num_sensors.argtypes = [C.c_void_p]
    lib.bfd_group_num_sensors.restype = C.c_int64
    lib.bfd_group_num_sensor_steps.argtypes = [C.c_void_p]
    lib.bfd_group_get_sensor_index.argtypes = [C.c_void_p, C.c_void_p]
    lib.bfd_group_get_sensors.argtypes = [C.c_void_p, C.c_void_p]
    lib.bfd_group_get_sensor_dft.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
    lib.bfd_group_get_map.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_int64, C.c_int64]
    lib.bfd_group_device_bytes.argtypes = [C.c_void_p]
    lib.bfd_group_device_bytes.restype = C.c_int64
    lib.bfd_group_peer_status.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    lib.bfd_placement_cache_release.argtypes = []
    lib.bfd_placement_cache_release.restype = C.c_int64
    if lib.bfd_abi_version() != 7:
        raise EngineError('libbabelfdtd_hip.so ABI version mismatch')
    _lib = lib
    return lib


def _check(rc, what):
    if rc != 0:
        raise EngineError('%s failed (rc=%d): %s' % (what, rc, load_library().bfd_last_error().decode()))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _estrides(a):
    return [s // a.itemsize for s in a.strides]


def mask_of(names):
    m = 0
    for n in names:
        if n not in MAP_BITS:
            raise ValueError('unknown map name %r (valid: %s)' % (n, ', '.join(MAP_BITS)))
        m |= 1 << MAP_BITS[n]
    return m


def ordered(names):
    return sorted(set(names), key=lambda n: MAP_BITS[n])


def list_devices():
    """[(ordinal, name)] -- counterpart of <backend>.ListDevices (SelFiles/SelFiles.py:245-262)."""
    lib = load_library()
    out = []
    for d in range(lib.bfd_device_count()):
        buf = C.create_string_buffer(256)
        _check(lib.bfd_device_name(d, buf, 256), 'bfd_device_name')
        out.append((d, buf.value.decode()))
    return out


def stable_dt(MaterialList, Frequency, QfactorCorrection, SpatialStep, AlphaCFL, QCorrection=1.0):
    lib = load_library()
    ml = np.ascontiguousarray(MaterialList, np.float64).reshape(-1, 5)
    qc = np.ascontiguousarray(np.broadcast_to(np.asarray(QCorrection, np.float64), (ml.shape[0],)))
    dt = lib.bfd_stable_dt(ml.shape[0], _ptr(ml), _ptr(qc), float(Frequency), int(bool(QfactorCorrection)),
                           float(SpatialStep), float(AlphaCFL))
    if dt <= 0:
        raise EngineError('bfd_stable_dt failed: ' + lib.bfd_last_error().decode())
    return dt


def material_tables(MaterialList, Frequency, QfactorCorrection, SpatialStep, dt, QCorrection=1.0):
    lib = load_library()
    ml = np.ascontiguousarray(MaterialList, np.float64).reshape(-1, 5)
    qc = np.ascontiguousarray(np.broadcast_to(np.asarray(QCorrection, np.float64), (ml.shape[0],)))
    t = np.zeros((7, ml.shape[0]), np.float32)
    c1k2 = np.zeros(2, np.float32)
    cmax = C.c_double()
    _check(lib.bfd_material_tables(ml.shape[0], _ptr(ml), _ptr(qc), float(Frequency), int(bool(QfactorCorrection)),
                                   float(SpatialStep), float(dt), _ptr(t), _ptr(c1k2), C.byref(cmax)),
           'bfd_material_tables')
    return t, c1k2, cmax.value


def dft_series(series, dt_sensor, freq, device=0):
    """Device single-bin DFT (+ peak) of a host (nSensor, nTs) float32 series: bfd_dft_series."""
    lib = load_library()
    x = np.ascontiguousarray(series, np.float32)
    F = np.zeros(x.shape[0], np.complex64)
    pk = np.zeros(x.shape[0], np.float32)
    _check(lib.bfd_dft_series(device, x.shape[0], x.shape[1], _ptr(x), float(dt_sensor), float(freq), _ptr(F.view(np.float32)), _ptr(pk)),
           'bfd_dft_series')
    return F, pk


class Engine:
    """One Z-slab of the domain on one GPU."""

    def __init__(self, N1, N2, N3, nMat, h, dt, freq, nt, k0=0, nk=None, NDelta=12, reflectionLimit=1e-5,
                 typeSource=0, sensorSub=1, sensorStart=0, selRMSorPeak=1, selMapsRMS=('Pressure',),
                 selMapsSensors=('Pressure',), qfactorCorrection=True, device=0, kernelVariant=0, rmsFirstStep=0, sensorMode=0):
        self.lib = load_library()
        if self.lib.bfd_device_count() <= 0:
            raise EngineError('no HIP device visible: the MI355X engine has no CPU fallback')
        nk = N3 - k0 if nk is None else nk
        self.selR = ordered(selMapsRMS)
        self.selS = ordered(selMapsSensors)
        self.cfg = Config(N1=N1, N2=N2, N3=N3, k0=k0, nk=nk, nMat=nMat, NDelta=NDelta, typeSource=typeSource,
                          sensorSub=sensorSub, sensorStart=sensorStart, nt=nt, selRMSorPeak=selRMSorPeak,
                          selMapsRMS=mask_of(self.selR), selMapsSensors=mask_of(self.selS),
                          qfactorCorrection=int(bool(qfactorCorrection)), device=device, kernelVariant=kernelVariant, rmsFirstStep=rmsFirstStep, sensorMode=sensorMode,
                          h=h, dt=dt, freq=freq, reflectionLimit=reflectionLimit)
        self.h = C.c_void_p()
        _check(self.lib.bfd_create(C.byref(self.cfg), C.byref(self.h)), 'bfd_create')
        self.shape = (N1, N2, nk)

    def close(self):
        if getattr(self, 'h', None) is not None and self.h:
            self.lib.bfd_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- inputs ----
    def set_stream(self, stream_ptr):
        _check(self.lib.bfd_set_stream(self.h, C.c_void_p(stream_ptr)), 'bfd_set_stream')

    def set_materials(self, MaterialList, QCorrection=1.0):
        ml = np.ascontiguousarray(MaterialList, np.float64).reshape(-1, 5)
        if ml.shape[0] != self.cfg.nMat:
            raise ValueError('MaterialList rows != nMat')
        qc = np.ascontiguousarray(np.broadcast_to(np.asarray(QCorrection, np.float64), (ml.shape[0],)))
        _check(self.lib.bfd_set_materials(self.h, _ptr(ml), _ptr(qc)), 'bfd_set_materials')

    def set_material_map(self, slab_with_ghosts, ghostLow, ghostHigh):
        """slab_with_ghosts: uint32 (N1,N2,ghostLow+nk+ghostHigh) view/array (any strides >= 0)."""
        a = np.asarray(slab_with_ghosts)
        if a.dtype != np.uint32:
            a = a.astype(np.uint32)
        assert a.shape == (self.shape[0], self.shape[1], self.shape[2] + ghostLow + ghostHigh), a.shape
        if any(s < 0 for s in a.strides):
            a = np.ascontiguousarray(a)
        s1, s2, s3 = _estrides(a)
        base = a.ctypes.data + ghostLow * a.strides[2]
        _check(self.lib.bfd_set_material_map(self.h, C.c_void_p(base), s1, s2, s3, ghostLow, ghostHigh),
               'bfd_set_material_map')

    def set_reflector(self, mask):
        if mask is None:
            _check(self.lib.bfd_set_reflector(self.h, None, 0, 0, 0), 'bfd_set_reflector')
            return
        a = self._dense_u32(mask)
        _check(self.lib.bfd_set_reflector(self.h, _ptr(a), *_estrides(a)), 'bfd_set_reflector')

    def _dense_u32(self, a):
        a = np.asarray(a)
        assert a.shape == self.shape, (a.shape, self.shape)
        if a.dtype != np.uint32 or any(s < 0 for s in a.strides):
            a = np.ascontiguousarray(a, np.uint32)
        # the ABI uploads the whole address span of the view: keep it dense
        if a.size and (sum((n - 1) * s for n, s in zip(a.shape, _estrides(a))) + 1) != a.size:
            a = np.ascontiguousarray(a)
        return a

    def set_sources(self, localIndex, row, wx, wy, wz, PulseSource):
        pulse = np.ascontiguousarray(np.atleast_2d(PulseSource), np.float64)
        self._pulse = pulse     # a large table is streamed from here in time tiles during the run: keep it alive with the engine
        li = np.asarray(localIndex)
        if li.size and (int(li.max()) >= 2 ** 32 or int(li.min()) < 0):
            raise ValueError('source index beyond 2^32 voxels: split the domain over devices (devices=[...])')
        li = np.ascontiguousarray(li, np.uint32)
        rw = np.ascontiguousarray(row, np.uint32)
        ws = [None if w is None else np.ascontiguousarray(w, np.float32) for w in (wx, wy, wz)]
        _check(self.lib.bfd_set_sources(self.h, li.size, _ptr(li), _ptr(rw), _ptr(ws[0]), _ptr(ws[1]), _ptr(ws[2]),
                                        _ptr(pulse), pulse.shape[0], pulse.shape[1]), 'bfd_set_sources')

    def set_sensor_map(self, sensor_slab):
        a = self._dense_u32(sensor_slab)
        n = C.c_int64()
        _check(self.lib.bfd_set_sensor_map(self.h, _ptr(a), *_estrides(a), C.byref(n)), 'bfd_set_sensor_map')
        return n.value

    # ---- stepping ----
    def run(self, nSteps):
        _check(self.lib.bfd_run(self.h, int(nSteps)), 'bfd_run')

    def half_step_stress(self, part=0, stream=None):
        """stream: raw HIP stream handle (int) to launch this part on, without synchronisation; None = the engine's stream."""
        if stream is None:
            _check(self.lib.bfd_half_step_stress_part(self.h, part), 'bfd_half_step_stress_part')
        else:
            _check(self.lib.bfd_half_step_stress_part_on(self.h, part, C.c_void_p(stream)), 'bfd_half_step_stress_part_on')

    def half_step_velocity(self, part=0, stream=None):
        if stream is None:
            _check(self.lib.bfd_half_step_velocity_part(self.h, part), 'bfd_half_step_velocity_part')
        else:
            _check(self.lib.bfd_half_step_velocity_part_on(self.h, part, C.c_void_p(stream)), 'bfd_half_step_velocity_part_on')

    def halo_fields(self):
        """Which fields of each halo group this slab reads from its Z-neighbours' planes:
        an all-fluid slab (tiled kernels, no solid tile) needs only Vz and Szz."""
        out = {}
        for g in (HALO_VELOCITY, HALO_STRESS):
            m = C.c_uint32()
            _check(self.lib.bfd_halo_fields(self.h, g, C.byref(m)), 'bfd_halo_fields')
            out[g] = [f for f in range(3) if m.value & (1 << f)]
        return out

    def sync(self):
        _check(self.lib.bfd_sync(self.h), 'bfd_sync')

    @property
    def step(self):
        return self.lib.bfd_current_step(self.h)

    def halo_region(self, group, f, side, send):
        p = C.c_void_p()
        n = C.c_size_t()
        _check(self.lib.bfd_halo_region(self.h, group, f, side, int(send), C.byref(p), C.byref(n)), 'bfd_halo_region')
        return p.value, n.value

    def timing_begin(self, perKernel=False):
        """perKernel: False/0 whole window only, True/1 + per half-step, 2 + per kernel class (timing_kernels)."""
        _check(self.lib.bfd_timing_begin(self.h, int(perKernel)), 'bfd_timing_begin')

    def timing_end(self):
        d = [C.c_double() for _ in range(4)]
        n = [C.c_int64() for _ in range(2)]
        _check(self.lib.bfd_timing_end(self.h, *[C.byref(x) for x in d], *[C.byref(x) for x in n]), 'bfd_timing_end')
        return {'total_ms': d[0].value, 'stress_ms': d[1].value, 'velocity_ms': d[2].value, 'other_ms': d[3].value,
                'n_stress': n[0].value, 'n_velocity': n[1].value}

    def timing_kernels(self):
        """After timing_begin(2) ... timing_end(): {class: (total ms, launches)} per kernel class."""
        ms = np.zeros(len(KERNEL_CLASSES), np.float64)
        n = np.zeros(len(KERNEL_CLASSES), np.int64)
        _check(self.lib.bfd_timing_kernels(self.h, _ptr(ms), _ptr(n)), 'bfd_timing_kernels')
        return {c: (float(ms[i]), int(n[i])) for i, c in enumerate(KERNEL_CLASSES)}

    def algorithmic_bytes(self, accumulating=True):
        """Algorithmic bytes per launch of each kernel class (bfd_algorithmic_bytes)."""
        b = np.zeros(len(KERNEL_CLASSES), np.float64)
        _check(self.lib.bfd_algorithmic_bytes(self.h, int(bool(accumulating)), _ptr(b)), 'bfd_algorithmic_bytes')
        return {c: float(b[i]) for i, c in enumerate(KERNEL_CLASSES)}

    def prepare(self):
        """Classes, run lists and the placement of the per-voxel arrays (bfd_prepare); before any halo() of a slab."""
        _check(self.lib.bfd_prepare(self.h), 'bfd_prepare')

    def reset(self):
        _check(self.lib.bfd_reset(self.h), 'bfd_reset')

    def set_placement(self, mode=1, search_limit_bytes=-1):
        """Placement policy of the per-voxel arrays (bfd_set_placement), before the first step: mode 0 = off; search_limit_bytes =
        throw-away memory the search for another memory region may hold (< 0: default rule, nothing on a shared device)."""
        _check(self.lib.bfd_set_placement(self.h, int(mode), int(search_limit_bytes)), 'bfd_set_placement')

    def placement_note(self):
        """What bfd_prepare found about the memory regions of the arrays and did about it."""
        return self.lib.bfd_placement_note(self.h).decode()

    # ---- outputs ----
    @property
    def num_sensors(self):
        return self.lib.bfd_num_sensors(self.h)

    @property
    def num_sensor_steps(self):
        return self.lib.bfd_num_sensor_steps(self.h)

    def sensor_index(self):
        idx = np.zeros(self.num_sensors, np.uint32)
        _check(self.lib.bfd_get_sensor_index(self.h, _ptr(idx)), 'bfd_get_sensor_index')
        return idx

    def sensors(self):
        out = np.zeros((len(self.selS), self.num_sensors, max(self.num_sensor_steps, 0)), np.float32)
        _check(self.lib.bfd_get_sensors(self.h, _ptr(out)), 'bfd_get_sensors')
        return out

    def sensor_dft(self, freq):
        """(F, peak): complex64 and float32 arrays [nSelSensors][nSensors] (bfd_get_sensor_dft)."""
        F = np.zeros((len(self.selS), self.num_sensors), np.complex64)
        pk = np.zeros((len(self.selS), self.num_sensors), np.float32)
        _check(self.lib.bfd_get_sensor_dft(self.h, float(freq), _ptr(F.view(np.float32)), _ptr(pk)), 'bfd_get_sensor_dft')
        return F, pk

    def get_map(self, kind, name, out=None):
        if out is None:
            out = np.zeros(self.shape, np.float32)
        _check(self.lib.bfd_get_map(self.h, kind, MAP_BITS[name], _ptr(out), *_estrides(out)), 'bfd_get_map')
        return out

    def get_field(self, name, out=None):
        if out is None:
            out = np.zeros(self.shape, np.float32)
        _check(self.lib.bfd_get_field(self.h, FIELD_NAMES.index(name), _ptr(out), *_estrides(out)), 'bfd_get_field')
        return out

    def tile_counts(self):
        n = [C.c_int32() for _ in range(5)]
        _check(self.lib.bfd_tile_counts(self.h, *[C.byref(x) for x in n]), 'bfd_tile_counts')
        lean = C.c_int32()
        _check(self.lib.bfd_tile_count_lean(self.h, C.byref(lean)), 'bfd_tile_count_lean')
        fused = C.c_int32()
        _check(self.lib.bfd_tile_count_fused(self.h, C.byref(fused)), 'bfd_tile_count_fused')
        return {'lossless_fluid': n[0].value, 'lossy_fluid': n[1].value, 'solid': n[2].value,
                'uniform_fluid': n[3].value, 'pml_fluid': n[4].value, 'lean_fluid': lean.value, 'fused_fluid': fused.value}

    def activity_counts(self):
        """(active, total) sub-tiles of the quiet-run map; (0, 0) when the engine works on every tile in every half-step."""
        a, t = C.c_int64(), C.c_int64()
        _check(self.lib.bfd_activity_counts(self.h, C.byref(a), C.byref(t)), 'bfd_activity_counts')
        return a.value, t.value

    @property
    def device_bytes(self):
        return self.lib.bfd_device_bytes(self.h)


class _SlabView(Engine):
    """Engine-shaped view of one slab of a Group (per-slab queries: tile counts, timing, raw fields). Owned by the group."""

    def __init__(self, lib, handle, cfg, shape, selR, selS):      # noqa: super().__init__ would create an engine
        self.lib, self.h, self.cfg, self.shape, self.selR, self.selS = lib, handle, cfg, shape, selR, selS

    def close(self):
        self.h = None

    def __del__(self):
        pass


PEER_PATHS = {0: 'same device (device copy)', 1: 'peer access both ways (direct hipMemcpyPeerAsync)', 2: 'STAGED through the host (peer access missing)'}


def decode_peer_status(codes, devices=None):
    """Status words of bfd_group_peer_status -> one dict per interface r | r+1: path, can_access / enabled in each direction."""
    out = []
    for r, c in enumerate(np.asarray(codes, np.int64).tolist()):
        e = {'interface': r, 'path': PEER_PATHS.get(c & 15, 'unknown'), 'direct': (c & 15) != 2,
             'can_access': [bool(c & 16), bool(c & 64)], 'enabled': [bool(c & 32), bool(c & 128)]}
        if devices is not None and r + 1 < len(devices):
            e['devices'] = [int(devices[r]), int(devices[r + 1])]
        out.append(e)
    return out


def placement_cache_release():
    """Frees the device buffers kept between solver calls for the placement of the arrays; returns the bytes freed."""
    return int(load_library().bfd_placement_cache_release())


class Group:
    """One solver call split into Z-slabs over several HIP devices of this process (bfd_group_*): whole-domain inputs and
    outputs, step loop and halo copies inside the library."""

    def __init__(self, devices, N1, N2, N3, nMat, h, dt, freq, nt, NDelta=12, reflectionLimit=1e-5, typeSource=0, sensorSub=1,
                 sensorStart=0, selRMSorPeak=1, selMapsRMS=('Pressure',), selMapsSensors=('Pressure',), qfactorCorrection=True,
                 kernelVariant=0, rmsFirstStep=0, sensorMode=0):
        self.lib = load_library()
        if self.lib.bfd_device_count() <= 0:
            raise EngineError('no HIP device visible: the MI355X engine has no CPU fallback')
        self.devices = [int(d) for d in devices]
        self.selR = ordered(selMapsRMS)
        self.selS = ordered(selMapsSensors)
        self.cfg = Config(N1=N1, N2=N2, N3=N3, k0=0, nk=N3, nMat=nMat, NDelta=NDelta, typeSource=typeSource,
                          sensorSub=sensorSub, sensorStart=sensorStart, nt=nt, selRMSorPeak=selRMSorPeak,
                          selMapsRMS=mask_of(self.selR), selMapsSensors=mask_of(self.selS),
                          qfactorCorrection=int(bool(qfactorCorrection)), device=self.devices[0], kernelVariant=kernelVariant,
                          rmsFirstStep=rmsFirstStep, sensorMode=sensorMode, h=h, dt=dt, freq=freq, reflectionLimit=reflectionLimit)
        self.h = C.c_void_p()
        dv = np.ascontiguousarray(self.devices, np.int32)
        _check(self.lib.bfd_group_create(C.byref(self.cfg), len(self.devices), _ptr(dv), C.byref(self.h)), 'bfd_group_create')
        self.shape = (N1, N2, N3)

    def close(self):
        if getattr(self, 'h', None) is not None and self.h:
            self.lib.bfd_group_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def size(self):
        return self.lib.bfd_group_size(self.h)

    def slab(self, r):
        """(k0, nk, device, engine view) of slab r."""
        k0, nk, dev, sim = C.c_int32(), C.c_int32(), C.c_int32(), C.c_void_p()
        _check(self.lib.bfd_group_slab(self.h, r, C.byref(k0), C.byref(nk), C.byref(dev), C.byref(sim)), 'bfd_group_slab')
        cfg = Config.from_buffer_copy(self.cfg)
        cfg.k0, cfg.nk, cfg.device = k0.value, nk.value, dev.value
        return k0.value, nk.value, dev.value, _SlabView(self.lib, sim, cfg, (self.shape[0], self.shape[1], nk.value), self.selR, self.selS)

    # ---- inputs (whole domain) ----
    def set_materials(self, MaterialList, QCorrection=1.0):
        ml = np.ascontiguousarray(MaterialList, np.float64).reshape(-1, 5)
        if ml.shape[0] != self.cfg.nMat:
            raise ValueError('MaterialList rows != nMat')
        qc = np.ascontiguousarray(np.broadcast_to(np.asarray(QCorrection, np.float64), (ml.shape[0],)))
        _check(self.lib.bfd_group_set_materials(self.h, _ptr(ml), _ptr(qc)), 'bfd_group_set_materials')

    def _u32(self, a):
        a = np.asarray(a)
        assert a.shape == self.shape, (a.shape, self.shape)
        if a.dtype != np.uint32 or any(s < 0 for s in a.strides):
            a = np.ascontiguousarray(a, np.uint32)
        return a

    def set_material_map(self, MaterialMap):
        a = self._u32(MaterialMap)
        _check(self.lib.bfd_group_set_material_map(self.h, _ptr(a), *_estrides(a)), 'bfd_group_set_material_map')

    def set_reflector(self, mask):
        if mask is None:
            _check(self.lib.bfd_group_set_reflector(self.h, None, 0, 0, 0), 'bfd_group_set_reflector')
            return
        a = self._u32(mask)
        _check(self.lib.bfd_group_set_reflector(self.h, _ptr(a), *_estrides(a)), 'bfd_group_set_reflector')

    def set_sources(self, globalIndex, row, wx, wy, wz, PulseSource):
        pulse = np.ascontiguousarray(np.atleast_2d(PulseSource), np.float64)
        self._pulse = pulse     # a large table is streamed from here during the run: keep it alive with the group
        gi = np.ascontiguousarray(globalIndex, np.int64)
        rw = np.ascontiguousarray(row, np.uint32)
        ws = [None if w is None else np.ascontiguousarray(w, np.float32) for w in (wx, wy, wz)]
        _check(self.lib.bfd_group_set_sources(self.h, gi.size, _ptr(gi), _ptr(rw), _ptr(ws[0]), _ptr(ws[1]), _ptr(ws[2]),
                                              _ptr(pulse), pulse.shape[0], pulse.shape[1]), 'bfd_group_set_sources')

    def set_sensor_map(self, SensorMap):
        a = self._u32(SensorMap)
        n = C.c_int64()
        _check(self.lib.bfd_group_set_sensor_map(self.h, _ptr(a), *_estrides(a), C.byref(n)), 'bfd_group_set_sensor_map')
        return n.value

    # ---- stepping ----
    def set_placement(self, mode=1, search_limit_bytes=-1):
        _check(self.lib.bfd_group_set_placement(self.h, int(mode), int(search_limit_bytes)), 'bfd_group_set_placement')

    def prepare(self):
        _check(self.lib.bfd_group_prepare(self.h), 'bfd_group_prepare')

    def run(self, nSteps):
        _check(self.lib.bfd_group_run(self.h, int(nSteps)), 'bfd_group_run')

    def sync(self):
        _check(self.lib.bfd_group_sync(self.h), 'bfd_group_sync')

    def reset(self):
        _check(self.lib.bfd_group_reset(self.h), 'bfd_group_reset')

    def timing_begin(self):
        _check(self.lib.bfd_group_timing_begin(self.h), 'bfd_group_timing_begin')

    def timing_end(self):
        d = [C.c_double() for _ in range(4)]
        ov = C.c_int32()
        _check(self.lib.bfd_group_timing_end(self.h, *[C.byref(x) for x in d], C.byref(ov)), 'bfd_group_timing_end')
        return {'total_ms': d[0].value, 'max_device_ms': d[1].value, 'host_issue_ms': d[2].value, 'halo_bytes_per_step': d[3].value,
                'overlapped': bool(ov.value), 'slabs': self.size, 'peer': self.peer_status()}

    def peer_status(self):
        """How the halo planes cross each interface (bfd_group_peer_status), decoded: one dict per interface."""
        n = max(self.size - 1, 0)
        codes = np.zeros(max(n, 1), np.int32)
        got = self.lib.bfd_group_peer_status(self.h, _ptr(codes), n)
        if got < 0:
            _check(got, 'bfd_group_peer_status')
        return decode_peer_status(codes[:n], self.devices)

    # ---- outputs (whole domain) ----
    @property
    def num_sensors(self):
        return self.lib.bfd_group_num_sensors(self.h)

    @property
    def num_sensor_steps(self):
        return self.lib.bfd_group_num_sensor_steps(self.h)

    def sensor_index(self):
        idx = np.zeros(self.num_sensors, np.uint32)
        _check(self.lib.bfd_group_get_sensor_index(self.h, _ptr(idx)), 'bfd_group_get_sensor_index')
        return idx

    def sensors(self):
        out = np.zeros((len(self.selS), self.num_sensors, max(self.num_sensor_steps, 0)), np.float32)
        _check(self.lib.bfd_group_get_sensors(self.h, _ptr(out)), 'bfd_group_get_sensors')
        return out

    def sensor_dft(self, freq):
        F = np.zeros((len(self.selS), self.num_sensors), np.complex64)
        pk = np.zeros((len(self.selS), self.num_sensors), np.float32)
        _check(self.lib.bfd_group_get_sensor_dft(self.h, float(freq), _ptr(F.view(np.float32)), _ptr(pk)), 'bfd_group_get_sensor_dft')
        return F, pk

    def get_map(self, kind, name, out=None):
        if out is None:
            out = np.zeros(self.shape, np.float32)
        _check(self.lib.bfd_group_get_map(self.h, kind, MAP_BITS[name], _ptr(out), *_estrides(out)), 'bfd_group_get_map')
        return out

    @property
    def device_bytes(self):
        return self.lib.bfd_group_device_bytes(self.h)

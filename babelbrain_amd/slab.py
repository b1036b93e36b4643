"""Z-slab domain decomposition: one process per GPU, neighbour halo exchange through
torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).

The reference has no multi-device path (every backend of its solver is single-device,
SURVEY.md 2.2); this is the new capability BASELINE.json asks for. In the solver's own linear
order i is fastest and k slowest (BabelIntegrationBASE.py:2508-2511), so a Z-slab is one
contiguous block and every halo plane is one contiguous N1*N2 run.

Launch with TORCH_NCCL_HIGH_PRIORITY=1 (bench.py sets it before creating the process group): the exchange kernel then
runs beside the interior kernels instead of queueing behind their workgroups.

Per time step and interface:
    exchange Vx,Vy,Vz boundary planes (2 per side)  -> stress half-step
    exchange Sxz,Syz,Szz boundary planes            -> velocity half-step
i.e. 2 x 6 fields x 2 planes x N1*N2 x 4 B = 96*N1*N2 bytes per interface per step, point to
point between neighbours only (open chain: rank r talks to r-1 and r+1, no collective).
"""
import numpy as np

from . import _engine
from ._engine import HALO_STRESS, HALO_VELOCITY, KIND_LAST, KIND_PEAK, KIND_RMS
from .PropagationModel import compact_sources, material_slab, n_steps, sensor_steps

MIN_PLANES = 4   # a slab must own at least the 2+2 planes its neighbours read


def partition(N3, world):
    """Balanced contiguous split of the N3 planes: [(k0, nk)] per rank."""
    if world < 1 or N3 < MIN_PLANES * world:
        raise ValueError('cannot split %d planes over %d ranks (>= %d planes per rank)' % (N3, world, MIN_PLANES))
    base, rem = divmod(N3, world)
    out, k0 = [], 0
    for r in range(world):
        nk = base + (1 if r < rem else 0)
        out.append((k0, nk))
        k0 += nk
    return out


class _DevBuf:
    """Exposes a raw device pointer through __cuda_array_interface__ so torch can alias it."""
    def __init__(self, ptr, nfloats):
        self.__cuda_array_interface__ = {'shape': (nfloats,), 'typestr': '<f4', 'data': (ptr, False), 'version': 2}


class HipSlab:
    """One slab on one MI355X; halo tensors alias the engine's own device memory (zero copy)."""
    @property
    def supports_parts(self):
        """Half-steps can run the boundary runs first, so the exchange overlaps the interior (kernelVariant 1 has no run
        lists: its part 1 is empty, an exchange started behind it would send the planes of the previous step)."""
        return self.eng.cfg.kernelVariant != 1

    def __init__(self, engine, device, host_staging=False):
        """host_staging=True: halos travel through pinned host tensors (for the gloo backend: a debugging
        aid that lets several ranks share one GPU; RCCL refuses that). Default: zero-copy device tensors."""
        import torch
        self.eng = engine
        self.torch = torch
        self.device = device
        self.host_staging = host_staging
        torch.cuda.set_device(device)
        # run the engine on torch's current stream so RCCL work orders against the kernels
        engine.set_stream(torch.cuda.current_stream(device).cuda_stream)
        self._t = {}
        for g in (HALO_VELOCITY, HALO_STRESS):
            for f in range(3):
                for side in (0, 1):
                    for send in (0, 1):
                        ptr, nbytes = engine.halo_region(g, f, side, send)
                        self._t[(g, f, side, send)] = torch.as_tensor(_DevBuf(ptr, nbytes // 4), device='cuda:%d' % device)
        self._h = {key: torch.empty(t.numel(), dtype=torch.float32).pin_memory() for key, t in self._t.items()} if host_staging else None

    def halo(self, group, f, side, send):
        return (self._h if self.host_staging else self._t)[(group, f, side, int(send))]

    def halo_fields(self):
        return self.eng.halo_fields()

    def before_send(self, group):
        if self.host_staging:
            for f in range(3):
                for side in (0, 1):
                    self._h[(group, f, side, 1)].copy_(self._t[(group, f, side, 1)])
            self.torch.cuda.synchronize(self.device)

    def after_recv(self, group, sides):
        if self.host_staging:
            for f in range(3):
                for side in sides:
                    self._t[(group, f, side, 0)].copy_(self._h[(group, f, side, 0)])

    def half_step_stress(self, part=0, stream=None):
        self.eng.half_step_stress(part, None if stream is None else stream.cuda_stream)

    def half_step_velocity(self, part=0, stream=None):
        self.eng.half_step_velocity(part, None if stream is None else stream.cuda_stream)

    def streams(self):
        """(main, side) torch streams for the overlapped step: the main one is the stream the engine runs on; the side
        stream carries the boundary part of a half-step and the halo exchange that waits for it. None with host staging."""
        if self.host_staging:
            return None
        if getattr(self, '_streams', None) is None:
            torch = self.torch
            # high priority: the small boundary kernels and the exchange must not queue behind the interior's workgroups
            self._streams = (torch.cuda.current_stream(self.device), torch.cuda.Stream(self.device, priority=-1))
        return self._streams

    def sync(self):
        self.torch.cuda.synchronize(self.device)

    def close(self):
        """Release the engine; the halo tensors alias its device memory, so they go first."""
        self._t = {}
        self._h = None
        self._streams = None
        self.eng.close()


ALL_FIELDS = {HALO_VELOCITY: [0, 1, 2], HALO_STRESS: [0, 1, 2]}


class SlabRunner:
    """Advances one slab in lock-step with its Z-neighbours.

    Every rank announces once which halo fields it reads (an all-fluid slab needs only Vz and Szz);
    across an interface each side sends exactly what the other side reads. If the slab engine can split
    a half-step (HipSlab), the boundary tiles run first, the exchange is started, and the interior tiles
    run while the planes travel."""

    def __init__(self, slab, rank, world, dist=None, group=None, overlap=None):
        self.slab, self.rank, self.world, self.group = slab, rank, world, group
        if world > 1 and dist is None:
            import torch.distributed as dist
        self.dist = dist
        self.low = rank - 1 if rank > 0 else None
        self.high = rank + 1 if rank < world - 1 else None
        self.bytes_sent = 0
        # Overlapped step (boundary part + exchange on a side stream beside the interior part) by default; a thin slab has
        # too little interior work to hide anything behind and pays for the extra launches, so it takes the blocking order
        # (BFD_OVERLAP_MIN_PLANES, default 64 planes; an explicit overlap= argument wins)
        import os
        thick = getattr(getattr(slab, 'eng', None), 'shape', (0, 0, 1 << 30))[2] >= int(os.environ.get('BFD_OVERLAP_MIN_PLANES', '64'))
        able = bool(getattr(slab, 'supports_parts', False))
        mine_overlap = (able and thick) if overlap is None else (bool(overlap) and able)
        if world > 1 and dist is not None:
            # the two orders exchange in a different sequence, so every rank must take the same one: every rank gathers every
            # rank's wish (whatever overlap= each was built with -- all ranks enter this collective) and the order is
            # overlapped only if all of them want and can
            flags = [None] * world
            dist.all_gather_object(flags, mine_overlap, group=group)
            mine_overlap = all(flags)
        self.overlap = mine_overlap and world > 1
        if self.overlap and world > 1:
            import warnings
            try:
                nccl = dist.get_backend(group) == 'nccl'
            except Exception:
                nccl = False
            if nccl and os.environ.get('TORCH_NCCL_HIGH_PRIORITY') != '1':
                # measured (rocprofv3 timeline): at default priority RCCL's kernel starts only when the
                # interior kernel has dispatched all of its workgroups, so the exchange is exposed
                warnings.warn('SlabRunner: set TORCH_NCCL_HIGH_PRIORITY=1 before init_process_group, otherwise the halo '
                              'exchange queues behind the interior kernels instead of running beside them')
        mine = slab.halo_fields() if hasattr(slab, 'halo_fields') else ALL_FIELDS
        self.needs = [mine]
        if world > 1:
            self.needs = [None] * world
            dist.all_gather_object(self.needs, mine, group=group)

    def _ops(self, halo_group):
        """The point-to-point operations of one exchange, built once per halo group (the tensors alias fixed device or
        host buffers, so the same descriptors serve every step)."""
        cache = self.__dict__.setdefault('_op_cache', {})
        if halo_group not in cache:
            dist, s = self.dist, self.slab
            ops, nbytes = [], 0
            # side 0 = my low face <-> peer's high ghosts, side 1 = my high face <-> peer's low ghosts
            for side, peer in ((0, self.low), (1, self.high)):
                if peer is None:
                    continue
                for f in self.needs[peer][halo_group]:          # what the peer reads from me
                    t = s.halo(halo_group, f, side, True)
                    ops.append(dist.P2POp(dist.isend, t, peer, self.group))
                    nbytes += t.numel() * 4
                for f in self.needs[self.rank][halo_group]:     # what I read from the peer
                    ops.append(dist.P2POp(dist.irecv, s.halo(halo_group, f, side, False), peer, self.group))
            cache[halo_group] = (ops, nbytes)
        return cache[halo_group]

    def bytes_per_step(self):
        """Bytes this rank sends per time step (both halo groups, both neighbours)."""
        if self.world == 1:
            return 0
        return sum(self._ops(g)[1] for g in (HALO_VELOCITY, HALO_STRESS))

    def exchange_start(self, halo_group):
        if self.world == 1:
            return []
        self.slab.before_send(halo_group)
        ops, nbytes = self._ops(halo_group)
        self.bytes_sent += nbytes
        return self.dist.batch_isend_irecv(ops) if ops else []

    def exchange_finish(self, reqs, halo_group):
        if self.world == 1:
            return
        for req in reqs:
            req.wait()
        self.slab.after_recv(halo_group, [sd for sd, p in ((0, self.low), (1, self.high)) if p is not None])

    def exchange(self, halo_group):
        self.exchange_finish(self.exchange_start(halo_group), halo_group)

    # --- overlapped half-step, device tensors (HipSlab): two streams -----------------------------------------------
    # main M: interior part (+ end-of-step work); side B: boundary part, then the exchange, which therefore waits for the
    # boundary part only. The host queues the interior (most of a millisecond of GPU work) BEFORE it spends ~0.1 ms
    # building the RCCL group, so neither that host time nor the transfer is exposed:
    #   B waits for M (everything before) -> part 1 on B -> (velocity only: M waits for part 1) -> part 2 on M
    #   -> on B: isend/irecv + wait -> M waits for B (the next half-step needs the received planes).
    def launch_parts(self, half, M, B):
        fn = self.slab.half_step_stress if half == HALO_STRESS else self.slab.half_step_velocity
        B.wait_stream(M)
        fn(1, stream=B)
        if half == HALO_VELOCITY:
            # the end-of-step work queued with velocity part 2 (sensors, non-Pressure accumulators) reads the whole slab,
            # boundary tiles included. The two parts of the stress half-step touch disjoint tiles and may run together.
            M.wait_event(B.record_event())
        fn(2, stream=M)

    def exchange_on(self, half, B):
        with self.slab.torch.cuda.stream(B):
            self.exchange_finish(self.exchange_start(half), half)

    def step(self):
        s = self.slab
        st = s.streams() if (self.overlap and hasattr(s, 'streams')) else None
        if st is not None:
            M, B = st
            for half in (HALO_STRESS, HALO_VELOCITY):      # the stress half-step produces the STRESS halo group, etc.
                self.launch_parts(half, M, B)
                self.exchange_on(half, B)
                M.wait_stream(B)
        elif self.overlap:
            # ghosts of V are current on entry (zero before the first step, exchanged at the end of every step)
            s.half_step_stress(1)
            w = self.exchange_start(HALO_STRESS)
            s.half_step_stress(2)
            self.exchange_finish(w, HALO_STRESS)
            s.half_step_velocity(1)
            w = self.exchange_start(HALO_VELOCITY)
            s.half_step_velocity(2)
            self.exchange_finish(w, HALO_VELOCITY)
        else:
            self.exchange(HALO_VELOCITY)
            s.half_step_stress()
            self.exchange(HALO_STRESS)
            s.half_step_velocity()

    def run(self, nSteps):
        for _ in range(nSteps):
            self.step()


def create_hip_slab(args, kwargs, rank, world, device, kernelVariant=0, local=None, host_staging=False, rmsFirstStep=0,
                    placement_search_bytes=None):
    """Build the engine for this rank's slab from the same arguments the reference passes to
    StaggeredFDTD_3D_with_relaxation (BASE:2338-2365). Returns (HipSlab, info).
    local=(N3, k0, nk, gl, gh): the volumes in `args` are already this rank's slab (MaterialMap with
    gl/gh ghost planes, Ox/Oy/Oz size-1) -- every rank built only its own share."""
    MaterialMap, MaterialList, Frequency, SourceMap, PulseSource, SpatialStep, Duration, SensorMap = args
    N1, N2 = MaterialMap.shape[:2]
    if local is None:
        N3 = MaterialMap.shape[2]
        k0, nk = partition(N3, world)[rank]
    else:
        N3, k0, nk, gl, gh = local
        if partition(N3, world)[rank] != (k0, nk):
            raise ValueError('local slab does not match the partition of N3 over the ranks')
    ml = np.ascontiguousarray(MaterialList, np.float64).reshape(-1, 5)
    DT = kwargs['DT']
    nt = n_steps(Duration, DT)
    one = np.array([1])
    eng = _engine.Engine(N1, N2, N3, ml.shape[0], SpatialStep, DT, Frequency, nt, k0=k0, nk=nk,
                         NDelta=kwargs.get('NDelta', 12), reflectionLimit=kwargs.get('ReflectionLimit', 1e-5),
                         typeSource=kwargs.get('TypeSource', 0), sensorSub=kwargs.get('SensorSubSampling', 1),
                         sensorStart=kwargs.get('SensorStart', 0), selRMSorPeak=kwargs.get('SelRMSorPeak', 1),
                         selMapsRMS=kwargs.get('SelMapsRMSPeakList', ('Pressure',)),
                         selMapsSensors=kwargs.get('SelMapsSensorsList', ('Pressure',)),
                         qfactorCorrection=kwargs.get('QfactorCorrection', True), device=device,
                         kernelVariant=kernelVariant, rmsFirstStep=rmsFirstStep)
    if placement_search_bytes is not None:      # a caller that owns the device may let the placement search further (bfd_set_placement)
        eng.set_placement(1, int(placement_search_bytes))
    eng.set_materials(ml, kwargs.get('QCorrection', 1.0))
    if local is None:
        view, gl, gh = material_slab(np.asarray(MaterialMap), k0, nk)
        ks = slice(k0, k0 + nk)
        lin, row, wx, wy, wz = compact_sources(np.asarray(SourceMap), kwargs.get('Ox', one), kwargs.get('Oy', one),
                                               kwargs.get('Oz', one), k0, nk)
    else:
        view = np.asarray(MaterialMap)
        ks = slice(0, nk)
        lin, row, wx, wy, wz = compact_sources(np.asarray(SourceMap), kwargs.get('Ox', one), kwargs.get('Oy', one),
                                               kwargs.get('Oz', one), 0, nk)
    eng.set_material_map(np.ascontiguousarray(view), gl, gh)
    if kwargs.get('ReflectorMask') is not None:
        eng.set_reflector(np.ascontiguousarray(kwargs['ReflectorMask'][:, :, ks]))
    eng.set_sources(lin, row, wx, wy, wz, PulseSource)
    eng.set_sensor_map(np.ascontiguousarray(np.asarray(SensorMap)[:, :, ks]))
    eng.prepare()            # classes, run lists, placement of the arrays: before the halo tensors alias them
    info = dict(k0=k0, nk=nk, nt=nt, DT=DT, N=(N1, N2, N3))
    return HipSlab(eng, device, host_staging=host_staging), info


def collect_slab_outputs(eng, kwargs, info):
    """This rank's share of the solver's return values (slab-local volumes, global sensor indices)."""
    sub, start = kwargs.get('SensorSubSampling', 1), kwargs.get('SensorStart', 0)
    Sensor = {'time': sensor_steps(info['nt'], sub, start) * info['DT']}
    sens = eng.sensors()
    for q, name in enumerate(eng.selS):
        Sensor[name] = sens[q]
    out = {'Sensor': Sensor, 'IndexSensorMap': eng.sensor_index(),
           'LastMap': {n: eng.get_map(KIND_LAST, n) for n in eng.selR}}
    mode = kwargs.get('SelRMSorPeak', 1)
    if mode & 1:
        out['RMS'] = {n: eng.get_map(KIND_RMS, n) for n in eng.selR}
    if mode & 2:
        out['Peak'] = {n: eng.get_map(KIND_PEAK, n) for n in eng.selR}
    return out


def merge_slab_outputs(parts):
    """Concatenate per-rank outputs (rank order) into whole-domain results: volumes along k,
    sensors along the sensor axis -- ascending global index, since slabs are ordered in k."""
    merged = {'Sensor': {'time': parts[0]['Sensor']['time']}}
    for key in parts[0]['Sensor']:
        if key != 'time':
            merged['Sensor'][key] = np.concatenate([p['Sensor'][key] for p in parts], axis=0)
    merged['IndexSensorMap'] = np.concatenate([p['IndexSensorMap'] for p in parts])
    for grp in ('LastMap', 'RMS', 'Peak'):
        if grp in parts[0]:
            merged[grp] = {n: np.concatenate([p[grp][n] for p in parts], axis=2) for n in parts[0][grp]}
    return merged

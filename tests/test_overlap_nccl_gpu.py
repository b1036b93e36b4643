"""The overlapped two-stream step of SlabRunner with REAL asynchronous RCCL traffic on one GPU: two or three slab
engines live in this process and exchange their halo planes through send-to-self pairs of a world-size-1 "nccl"
group (the same tensors, aliasing engine memory, the multi-GPU path sends). Stream choreography = SlabRunner's own
(launch_parts: boundary part on the side stream, interior on the main one; the exchange rides the side stream).
The decomposed run must equal the single-domain run bit for bit -- a missing dependency shows up as a mismatch."""
import os

import numpy as np
import pytest

from babelbrain_amd import harness as H
from babelbrain_amd import slab
from babelbrain_amd._engine import HALO_STRESS, HALO_VELOCITY
from tests.util import oracle_dt

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def self_group():
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip('a process group already exists in this process')
    os.environ['TORCH_NCCL_HIGH_PRIORITY'] = '1'        # as bench.py does: the exchange really runs beside the interior kernels
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29577')
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize('config,world,nz', [('C2', 2, 160), ('C3', 3, 200)])
def test_two_stream_overlap_with_rccl_self_sends(self_group, config, world, nz):
    import torch
    from babelbrain_amd import PropagationModel
    dist = self_group
    a, k, info = H.make_problem(config, N=(64, 56, nz), steps=150, stable_dt_fn=oracle_dt)
    k['SelMapsRMSPeakList'] = ['Pressure', 'Vz']
    k['SelMapsSensorsList'] = ['Pressure', 'Vx']
    k['SelRMSorPeak'] = 3
    ref = PropagationModel(kernelVariant=3).StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    slabs, infos, runners = [], [], []
    for r in range(world):
        s, i = slab.create_hip_slab(a, k, r, world, 0)
        slabs.append(s); infos.append(i)
        runners.append(slab.SlabRunner(s, 0, 1))            # only its launch_parts() is used here
    M, B = slabs[0].streams()
    needs = [s.halo_fields() for s in slabs]
    for _ in range(info['nt']):
        for half in (HALO_STRESS, HALO_VELOCITY):
            for rn in runners:
                rn.launch_parts(half, M, B)
            with torch.cuda.stream(B):
                ops = []
                for r in range(world - 1):                    # send-to-self pairs are matched in posting order
                    lo, hi = slabs[r], slabs[r + 1]
                    for f in needs[r + 1][half]:
                        ops += [dist.P2POp(dist.isend, lo.halo(half, f, 1, True), 0), dist.P2POp(dist.irecv, hi.halo(half, f, 0, False), 0)]
                    for f in needs[r][half]:
                        ops += [dist.P2POp(dist.isend, hi.halo(half, f, 0, True), 0), dist.P2POp(dist.irecv, lo.halo(half, f, 1, False), 0)]
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
            M.wait_stream(B)
    torch.cuda.synchronize()
    merged = slab.merge_slab_outputs([slab.collect_slab_outputs(s.eng, k, i) for s, i in zip(slabs, infos)])
    Sensor, Last, RMS, Peak, Inp = ref
    assert np.array_equal(merged['IndexSensorMap'], Inp['IndexSensorMap'])
    for n in ('Pressure', 'Vx'):
        assert np.array_equal(merged['Sensor'][n], Sensor[n]), n
    for n in RMS:
        assert np.array_equal(merged['RMS'][n], RMS[n]), n
        assert np.array_equal(merged['Peak'][n], Peak[n]), n
        assert np.array_equal(merged['LastMap'][n], Last[n]), n
    assert RMS['Pressure'].max() > 0
    for s in slabs:
        s.eng.close()

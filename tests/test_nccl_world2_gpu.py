"""Two ranks on two different GPUs exchanging halos over RCCL (xGMI): the first real multi-GPU run of the slab path
is a test, not the benchmark. Skipped on boxes with fewer than two GPUs (the gpurun pool has one per box; the driver's
8-GPU node runs it). Both step orders (two-stream overlapped, blocking) must reproduce the single-domain run bit for bit."""
import os

import numpy as np
import pytest

from babelbrain_amd import harness as H
from tests.util import oracle_dt, run_ranks

pytestmark = pytest.mark.gpu


def _problem():
    a, k, info = H.make_problem('C2', N=(64, 56, 160), steps=150, stable_dt_fn=oracle_dt)
    k['SelMapsRMSPeakList'] = ['Pressure', 'Vz']
    k['SelRMSorPeak'] = 3
    return a, k, info


def _worker(rank, world, port, q, overlap):
    import torch
    import torch.distributed as dist
    from babelbrain_amd import slab
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('TORCH_NCCL_HIGH_PRIORITY', '1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    torch.cuda.set_device(rank)
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', rank))
    try:
        a, k, info = _problem()
        s, sinfo = slab.create_hip_slab(a, k, rank, world, rank, kernelVariant=3)
        runner = slab.SlabRunner(s, rank, world, dist, overlap=overlap)
        assert runner.overlap == overlap
        runner.run(info['nt'])
        torch.cuda.synchronize()
        parts = [None] * world
        dist.all_gather_object(parts, slab.collect_slab_outputs(s.eng, k, sinfo))
        if rank == 0:
            q.put(slab.merge_slab_outputs(parts))
        dist.barrier()
        s.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize('overlap', [True, False])
def test_two_gpus_over_rccl_match_single_domain(overlap):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs (RCCL refuses two ranks on one device)')
    from babelbrain_amd import PropagationModel
    merged = run_ranks(_worker, 2, timeout=800, extra=(overlap,))
    a, k, info = _problem()
    Sensor, Last, RMS, Peak, Inp = PropagationModel(kernelVariant=3).StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    assert np.array_equal(merged['IndexSensorMap'], Inp['IndexSensorMap'])
    assert np.array_equal(merged['Sensor']['Pressure'], Sensor['Pressure'])
    for n in RMS:
        assert np.array_equal(merged['RMS'][n], RMS[n]) and np.array_equal(merged['Peak'][n], Peak[n])
    assert RMS['Pressure'].max() > 0

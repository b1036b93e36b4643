"""Two and three processes sharing GPU 0 (gloo backend, halos staged through pinned host memory) run the
real SlabRunner code of babelbrain_amd/slab.py around HIP slab engines; the merged outputs must equal the
single-domain HIP run exactly. RCCL cannot be exercised on a one-GPU box (it refuses two ranks on one
device); everything else of the multi-rank path is."""
import os

import numpy as np
import pytest

from babelbrain_amd import harness as H
from tests.util import oracle_dt, run_ranks

pytestmark = pytest.mark.gpu


def _problem():
    a, k, info = H.make_problem('C2', N=(64, 56, 120), steps=140, stable_dt_fn=oracle_dt)
    k['SelMapsRMSPeakList'] = ['Pressure', 'Vz']
    k['SelRMSorPeak'] = 3
    return a, k, info


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from babelbrain_amd import slab
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        a, k, info = _problem()
        s, sinfo = slab.create_hip_slab(a, k, rank, world, 0, kernelVariant=3, host_staging=True)
        runner = slab.SlabRunner(s, rank, world, dist, overlap=(world == 3))    # both step orders
        runner.run(info['nt'])
        torch.cuda.synchronize()
        parts = [None] * world
        dist.all_gather_object(parts, slab.collect_slab_outputs(s.eng, k, sinfo))
        if rank == 0:
            q.put(slab.merge_slab_outputs(parts))
        dist.barrier()
        s.eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize('world', [2, 3])
def test_multiprocess_slabs_match_single_domain(world):
    from babelbrain_amd import PropagationModel
    merged = run_ranks(_worker, world, timeout=800)
    a, k, info = _problem()
    Sensor, Last, RMS, Peak, Inp = PropagationModel(kernelVariant=3).StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    assert np.array_equal(merged['IndexSensorMap'], Inp['IndexSensorMap'])
    assert np.array_equal(merged['Sensor']['Pressure'], Sensor['Pressure'])
    for n in RMS:
        assert np.array_equal(merged['RMS'][n], RMS[n]) and np.array_equal(merged['Peak'][n], Peak[n])
    assert RMS['Pressure'].max() > 0

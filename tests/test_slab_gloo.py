"""N>1 path on CPU: two processes, gloo backend, the Z-slab runner of babelbrain_amd/slab.py driving
the oracle's slab form. The decomposed run must reproduce the single-domain run exactly."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from babelbrain_amd import harness as H
from babelbrain_amd import slab
from oracle import oracle as O
from tests.util import oracle_dt, run_ranks


def _problem():
    a, k, info = H.make_problem('C2', N=(40, 36, 64), steps=70, stable_dt_fn=oracle_dt)
    k['SelMapsRMSPeakList'] = ['Pressure', 'Vz', 'Sigmaxz']
    k['SelMapsSensorsList'] = ['Pressure', 'Vx']
    k['SelRMSorPeak'] = 3
    refl = np.zeros(a[0].shape, np.uint32)
    refl[18:22, 16:20, 30:35] = 1             # straddles the slab interface at k=32
    k['ReflectorMask'] = refl
    return a, k, info


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        a, k, info = _problem()
        k0, nk = slab.partition(a[0].shape[2], world)[rank]
        s = O.OracleSlab(a, k, k0, nk, nthreads=2)
        runner = slab.SlabRunner(s, rank, world, dist)
        runner.run(info['nt'])
        parts = [None] * world
        dist.all_gather_object(parts, s.outputs())
        if rank == 0:
            q.put((slab.merge_slab_outputs(parts), runner.bytes_sent))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize('world', [2, 3])
def test_slab_decomposition_matches_single_domain(world):
    merged, sent = run_ranks(_worker, world, timeout=500)
    a, k, info = _problem()
    ref = O.StaggeredFDTD_3D_with_relaxation(*a, **k)
    Sensor, Last, RMS, Peak, Inp = ref
    assert np.array_equal(merged['IndexSensorMap'], Inp['IndexSensorMap'])
    for n in Sensor:
        assert np.array_equal(merged['Sensor'][n], Sensor[n]), n
    for n in RMS:
        assert np.array_equal(merged['RMS'][n], RMS[n]), n
        assert np.array_equal(merged['Peak'][n], Peak[n]), n
        assert np.array_equal(merged['LastMap'][n], Last[n]), n
    assert RMS['Pressure'].max() > 0
    # 96*N1*N2 bytes per interface per step (both directions) => 48*N1*N2 sent by rank 0 per step
    N1, N2 = a[0].shape[:2]
    assert sent == 48 * N1 * N2 * info['nt']


def test_partition():
    assert slab.partition(512, 8) == [(64 * r, 64) for r in range(8)]
    p = slab.partition(70, 3)
    assert sum(n for _, n in p) == 70 and p[0][0] == 0 and all(p[i][0] + p[i][1] == p[i + 1][0] for i in range(2))
    with pytest.raises(ValueError):
        slab.partition(10, 4)

"""Z-slab decomposition of the HIP engine on ONE GPU: several slab engines live on the same device
and exchange their halo planes by device-to-device copies of the very tensors the RCCL path sends
(babelbrain_amd/slab.py). The decomposed run must equal the single-domain run bit for bit."""
import numpy as np
import pytest

from babelbrain_amd import harness as H
from babelbrain_amd import slab
from babelbrain_amd._engine import HALO_STRESS, HALO_VELOCITY
from tests.util import oracle_dt

pytestmark = pytest.mark.gpu


def _exchange(slabs, group):
    """What SlabRunner.exchange does over RCCL, as device copies: each side receives the fields it reads."""
    for r in range(len(slabs) - 1):
        lo, hi = slabs[r], slabs[r + 1]
        for f in hi.halo_fields()[group]:
            hi.halo(group, f, 0, False).copy_(lo.halo(group, f, 1, True))    # low slab's top planes -> high slab's low ghosts
        for f in lo.halo_fields()[group]:
            lo.halo(group, f, 1, False).copy_(hi.halo(group, f, 0, True))


@pytest.mark.parametrize('config,variant,world,split', [('C2', 3, 2, False), ('C3', 3, 3, True), ('C2', 2, 2, True), ('C1', 3, 2, False),
                                                        ('C2', 3, 3, True), ('C1', 1, 2, True)])
def test_slabs_on_one_gpu_match_single_domain(config, variant, world, split):
    import torch
    from babelbrain_amd import PropagationModel
    a, k, info = H.make_problem(config, N=(64, 56, 200 if split else 96), steps=130, stable_dt_fn=oracle_dt)
    k['SelMapsRMSPeakList'] = ['Pressure', 'Vz', 'Sigmazz', 'Sigmaxx']
    k['SelMapsSensorsList'] = ['Pressure', 'Sigmayy']
    k['SelRMSorPeak'] = 3
    ref = PropagationModel(kernelVariant=variant).StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    slabs, infos = [], []
    for r in range(world):
        s, i = slab.create_hip_slab(a, k, r, world, 0, kernelVariant=variant)
        slabs.append(s); infos.append(i)
    for _ in range(info['nt']):
        if split:      # SlabRunner's overlapped order: boundary tiles, exchange, interior tiles
            for s in slabs:
                s.half_step_stress(1)
            _exchange(slabs, HALO_STRESS)
            for s in slabs:
                s.half_step_stress(2)
            for s in slabs:
                s.half_step_velocity(1)
            _exchange(slabs, HALO_VELOCITY)
            for s in slabs:
                s.half_step_velocity(2)
        else:
            _exchange(slabs, HALO_VELOCITY)
            for s in slabs:
                s.half_step_stress()
            _exchange(slabs, HALO_STRESS)
            for s in slabs:
                s.half_step_velocity()
    torch.cuda.synchronize()
    merged = slab.merge_slab_outputs([slab.collect_slab_outputs(s.eng, k, i) for s, i in zip(slabs, infos)])
    Sensor, Last, RMS, Peak, Inp = ref
    assert np.array_equal(merged['IndexSensorMap'], Inp['IndexSensorMap'])
    for n in ('Pressure', 'Sigmayy'):
        assert np.array_equal(merged['Sensor'][n], Sensor[n]), n
    for n in RMS:
        assert np.array_equal(merged['RMS'][n], RMS[n]), n
        assert np.array_equal(merged['Peak'][n], Peak[n]), n
        assert np.array_equal(merged['LastMap'][n], Last[n]), n
    assert RMS['Pressure'].max() > 0
    for s in slabs:
        s.eng.close()


def test_collapsed_all_fluid_slabs_with_reduced_halo():
    """All-fluid slabs keep only Szz and exchange only Vz / Szz planes; source plane and sensors sit in
    different slabs."""
    import torch
    from babelbrain_amd import PropagationModel
    a, k, info = H.make_problem('C3', N=(64, 48, 160), steps=140, stable_dt_fn=oracle_dt)
    ref = PropagationModel(kernelVariant=3).StaggeredFDTD_3D_with_relaxation(*a, SILENT=True, **k)
    world = 3
    slabs, infos = zip(*[slab.create_hip_slab(a, k, r, world, 0, kernelVariant=3) for r in range(world)])
    assert all(s.halo_fields() == {HALO_VELOCITY: [2], HALO_STRESS: [2]} for s in slabs)
    for _ in range(info['nt']):
        for s in slabs:
            s.half_step_stress(1)
        _exchange(slabs, HALO_STRESS)
        for s in slabs:
            s.half_step_stress(2)
        for s in slabs:
            s.half_step_velocity(1)
        _exchange(slabs, HALO_VELOCITY)
        for s in slabs:
            s.half_step_velocity(2)
    torch.cuda.synchronize()
    merged = slab.merge_slab_outputs([slab.collect_slab_outputs(s.eng, k, i) for s, i in zip(slabs, infos)])
    assert np.array_equal(merged['Sensor']['Pressure'], ref[0]['Pressure'])
    assert np.array_equal(merged['RMS']['Pressure'], ref[2]['Pressure']) and ref[2]['Pressure'].max() > 0
    for s in slabs:
        s.eng.close()

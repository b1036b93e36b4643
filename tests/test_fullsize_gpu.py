"""BASELINE.json's full single-GPU size (C3, 512^3) through size-independent properties, beside the direct comparison
with the oracle at that size (tests/test_configs_gpu.py::test_c3_full_size_against_oracle):
  * the three independent device implementations (one-thread-per-voxel variant 1, dense LDS-tiled variant 2,
    class-specialised variant 3) must agree exactly on the same inputs;
  * linearity: a source scaled by 2 scales every output by 2 (to rounding: the flushed-denormal arithmetic
    makes it exact only up to an ulp here and there, in the oracle as well);
  * a Z-slab decomposed run equals the single-domain run.
The small-grid tests tie variant 1..3 to the oracle; these tie the full-size run to them."""
import numpy as np
import pytest

from babelbrain_amd import harness as H
from babelbrain_amd import slab
from babelbrain_amd._engine import HALO_STRESS, HALO_VELOCITY

pytestmark = pytest.mark.gpu
STEPS = 140


@pytest.fixture(scope='module')
def c3():
    from babelbrain_amd import _engine, RayleighAndBHTE
    a, k, info = H.make_problem('C3', steps=STEPS, full_sensors=False, forward=RayleighAndBHTE.ForwardSimple,
                                stable_dt_fn=lambda ml, f, h, c: _engine.stable_dt(ml, f, True, h, c))
    assert a[0].shape == (512, 512, 512)
    return a, k, info


def _run(a, k, variant, scale=1.0):
    from babelbrain_amd import PropagationModel
    args = list(a)
    if scale != 1.0:
        args[4] = a[4] * scale
    out = PropagationModel(kernelVariant=variant).StaggeredFDTD_3D_with_relaxation(*args, SILENT=True, **k)
    return out[0]['Pressure'], out[2]['Pressure']


def test_variants_agree_and_linearity(c3):
    a, k, info = c3
    s3, r3 = _run(a, k, 3)
    assert r3.max() > 0 and np.count_nonzero(r3) > 1e6
    for v in (1, 2):
        sv, rv = _run(a, k, v)
        assert np.array_equal(sv, s3), 'sensors differ between variant %d and 3' % v
        assert np.array_equal(rv, r3), 'RMS map differs between variant %d and 3' % v
    s2x, r2x = _run(a, k, 3, scale=2.0)
    from tests.util import rel_l2
    assert rel_l2(s2x, 2 * s3) < 1e-6 and rel_l2(r2x, 2 * r3) < 1e-6


def test_slabs_equal_single_domain_at_full_size(c3):
    import torch
    a, k, info = c3
    _, r3 = _run(a, k, 3)
    world = 2
    slabs, infos = zip(*[slab.create_hip_slab(a, k, r, world, 0, kernelVariant=3) for r in range(world)])

    def exchange(group):
        lo, hi = slabs
        for f in hi.halo_fields()[group]:
            hi.halo(group, f, 0, False).copy_(lo.halo(group, f, 1, True))
        for f in lo.halo_fields()[group]:
            lo.halo(group, f, 1, False).copy_(hi.halo(group, f, 0, True))
    for _ in range(info['nt']):
        for s in slabs:
            s.half_step_stress(1)
        exchange(HALO_STRESS)
        for s in slabs:
            s.half_step_stress(2)
        for s in slabs:
            s.half_step_velocity(1)
        exchange(HALO_VELOCITY)
        for s in slabs:
            s.half_step_velocity(2)
    torch.cuda.synchronize()
    merged = slab.merge_slab_outputs([slab.collect_slab_outputs(s.eng, k, i) for s, i in zip(slabs, infos)])
    assert np.array_equal(merged['RMS']['Pressure'], r3)
    for s in slabs:
        s.eng.close()

"""HIP Rayleigh integral (bfd_rayleigh_forward through babelbrain_amd.RayleighAndBHTE.ForwardSimple)
against the float64 numpy oracle. Tolerance 1e-5 relative L2 (the north-star bar for field outputs):
geometry and phase reduction are float64 on the device, so the observed error is ~1e-7."""
import numpy as np
import pytest

from babelbrain_amd import harness as H
from oracle import rayleigh_oracle as RO
from tests.util import rel_l2

pytestmark = pytest.mark.gpu


def _case(M_rings, N, seed, kimag=0.0):
    from babelbrain_amd import RayleighAndBHTE as R
    rng = np.random.default_rng(seed)
    pts, ds = H._bowl_points(60e-3, 55e-3, M_rings, 0.0)
    u0 = (rng.normal(size=len(ds)) + 1j * rng.normal(size=len(ds))).astype(np.complex64)
    rf = np.stack([rng.uniform(-40e-3, 40e-3, N), rng.uniform(-40e-3, 40e-3, N), rng.uniform(20e-3, 160e-3, N)], 1).astype(np.float32)
    k = np.array(2 * np.pi * 700e3 / 1500.0 + 1j * kimag).astype(np.complex64)
    got = R.ForwardSimple(k, pts.astype(np.float32), ds.astype(np.float32), u0, rf)
    ref = RO.ForwardSimple(k, pts.astype(np.float32), ds.astype(np.float32), u0, rf)
    return got, ref, R.last_kernel_ms, len(ds)


@pytest.mark.parametrize('rings,N,kimag', [(12, 5000, 0.0), (30, 20001, 0.0), (20, 3000, -4.5), (1, 7, 0.0)])
def test_forward_simple_matches_oracle(rings, N, kimag):
    got, ref, ms, M = _case(rings, N, 1, kimag)
    assert got.dtype == np.complex64 and got.shape == (N,)
    e = max(rel_l2(got.real, ref.real), rel_l2(got.imag, ref.imag))
    assert e <= 1e-5, e
    print('M=%d N=%d: rel L2 %.2e, %.3f ms, %.2f Gpairs/s' % (M, N, e, ms, M * N / ms / 1e6))


def test_source_plane_matches_harness():
    """The bench's source plane (numpy, harness.rayleigh_plane) equals the device evaluation."""
    from babelbrain_amd import RayleighAndBHTE as R
    h = H.spatial_step(500e3, 6)
    xs = (np.arange(48) - 23.5) * h
    pts, ds = H._bowl_points(50e-3, 50e-3, 16, 0.0)
    u0 = np.ones(len(ds), np.complex64)
    z = pts[:, 2].max() + 2 * h
    ref = H.rayleigh_plane(pts, ds, u0.astype(np.complex128), 500e3, 1500.0, xs, xs, z)
    X, Y = np.meshgrid(xs, xs, indexing='ij')
    rf = np.stack([X.ravel(), Y.ravel(), np.full(X.size, z)], 1)
    got = R.ForwardSimple(2 * np.pi * 500e3 / 1500.0, pts, ds, u0, rf).reshape(X.shape)
    assert rel_l2(np.abs(got), np.abs(ref)) < 1e-5
    assert np.abs(np.angle(got * np.conj(ref))).max() < 1e-4


def test_empty_and_errors():
    from babelbrain_amd import RayleighAndBHTE as R
    out = R.ForwardSimple(1000.0, np.zeros((3, 3), np.float32), np.ones(3, np.float32), np.ones(3, np.complex64), np.zeros((0, 3), np.float32))
    assert out.shape == (0,)
    with pytest.raises(ValueError):
        R.ForwardSimple(1000.0, np.zeros((3, 3), np.float32), np.ones(2, np.float32), np.ones(3, np.complex64), np.zeros((4, 3), np.float32))
    assert abs(R.SpeedofSoundWater(20.0) - 1482.36) < 0.05


def test_field_points_shared_among_devices(monkeypatch):
    """set_devices([...]) / BABELFDTD_DEVICES: the field points are dealt to the listed devices in contiguous shares, one host
    thread each (an ordinal may repeat: here all shares run on device 0, on a multi-GPU node every visible device takes one).
    The result is the single-device one bit for bit (a point's sum does not depend on its neighbours in the launch)."""
    import torch
    from babelbrain_amd import RayleighAndBHTE as R
    rng = np.random.default_rng(3)
    pts, ds = H._bowl_points(60e-3, 55e-3, 14, 0.0)
    u0 = (rng.normal(size=len(ds)) + 1j * rng.normal(size=len(ds))).astype(np.complex64)
    N = 10007
    rf = np.stack([rng.uniform(-40e-3, 40e-3, N), rng.uniform(-40e-3, 40e-3, N), rng.uniform(20e-3, 160e-3, N)], 1).astype(np.float32)
    k = 2 * np.pi * 700e3 / 1500.0
    R.set_devices(None)
    monkeypatch.delenv('BABELFDTD_DEVICES', raising=False)
    one = R.ForwardSimple(k, pts, ds, u0, rf)
    try:
        R.set_devices([0, 0, 0])
        assert np.array_equal(R.ForwardSimple(k, pts, ds, u0, rf), one)
        assert np.array_equal(R.ForwardSimple(k, pts, ds, u0, rf[:50]), one[:50])           # too few points to share: one device
        R.set_devices('all')
        assert np.array_equal(R.ForwardSimple(k, pts, ds, u0, rf), one)
        nd = torch.cuda.device_count()
        R.set_devices(list(range(nd)) + [nd])                                               # one ordinal too many: refused, not skipped
        with pytest.raises(Exception):
            R.ForwardSimple(k, pts, ds, u0, rf)
    finally:
        R.set_devices(None)
    monkeypatch.setenv('BABELFDTD_DEVICES', '0,0')
    try:
        assert np.array_equal(R.ForwardSimple(k, pts, ds, u0, rf), one)
    finally:
        R.set_devices(None)

"""HDF5 dictionary files (`SaveToH5py` / `ReadFromH5py`): layout held to a file the reference ships
(TranscranialModeling/MapPichardo.h5, structure captured by tests/golden/make_golden.py), round trips of
every value kind a DataForSim dictionary carries, and the DataForSim built from golden inputs."""
import os

import numpy as np
import pytest

from babelbrain_amd import datafile as DF

try:
    DF.backend()
except ImportError:                                      # no h5py and no libhdf5 on this machine
    pytest.skip('no HDF5 library available', allow_module_level=True)


def _example():
    rng = np.random.default_rng(5)
    return {'p_amp': rng.uniform(0, 1, (5, 6, 7)).astype(np.float32),
            'p_complex': (rng.normal(size=(5, 6, 7)) + 1j * rng.normal(size=(5, 6, 7))).astype(np.complex64),
            'MaterialMap': rng.integers(0, 5, (5, 6, 7)).astype(np.uint32), 'AirMask': rng.integers(0, 2, (5, 6, 7)).astype(np.uint8),
            'Material': np.array([[1000.0, 1500.0, 0, 0, 0], [1896.5, 2476.0, 1542.0, 81.0, 164.0]]),
            'x_vec': np.linspace(-1e-2, 1e-2, 5), 'SpatialStep': 3.675e-4, 'TargetLocation': np.array([2, 3, 4]),
            'bDoRefocusing': True, 'ZIntoSkinPixels': 3, 'affine': np.eye(4), 'name': 'Tx Ø 50', 'nothing': None,
            'mask': np.array([True, False, True]), 'empty': np.zeros((0, 3)), 'c128': np.array([1 + 2j]),
            'nested': {'a': np.arange(3, dtype=np.int16), 'l': [1, 2.5, 'x', np.ones(2)], 't': (1, 2), 'deep': {'v': np.float32(2.5)}}}


def _same(a, b):
    if isinstance(a, dict):
        return isinstance(b, dict) and sorted(a) == sorted(b) and all(_same(a[k], b[k]) for k in a)
    if isinstance(a, (list, tuple)):
        return type(a) is type(b) and len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    if isinstance(a, np.ndarray):
        return isinstance(b, np.ndarray) and a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b)
    if a is None:
        return b is None
    return type(b) in (bool, int, float, complex, str) and a == b


def test_round_trip_of_every_value_kind(tmp_path):
    d = _example()
    fn = str(tmp_path / 'DataForSim.h5')
    DF.SaveToH5py(d, fn)
    r = DF.ReadFromH5py(fn)
    assert _same(d, r)
    # inputs are not modified and a second save truncates
    DF.SaveToH5py({'only': np.arange(4)}, fn)
    assert list(DF.ReadFromH5py(fn)) == ['only']


def test_ndarray_layout_matches_the_reference_file(golden, tmp_path):
    """Every ndarray = one dataset of the array's dtype/shape + scalar attribute type="ndarray" stored as a
    variable-length UTF-8 string: exactly what the reference's own MapPichardo.h5 looks like."""
    g, meta = golden
    want = meta['h5pysimple_layout']
    arrays = {k: np.zeros(v['shape'], v['dtype']) for k, v in want.items()}
    fn = str(tmp_path / 'like_pichardo.h5')
    DF.SaveToH5py(arrays, fn, use_h5py=False)
    assert DF.describe(fn) == want


def test_errors(tmp_path):
    with pytest.raises(FileNotFoundError):
        DF.ReadFromH5py(str(tmp_path / 'absent.h5'))
    bad = tmp_path / 'bad.h5'
    bad.write_bytes(b'not hdf5')
    with pytest.raises(IOError):
        DF.ReadFromH5py(str(bad), use_h5py=False)
    with pytest.raises(TypeError):
        DF.SaveToH5py({'f': object()}, str(tmp_path / 'x.h5'))
    with pytest.raises(TypeError):
        DF.SaveToH5py({'a/b': 1}, str(tmp_path / 'x.h5'))
    with pytest.raises(TypeError):
        DF.SaveToH5py([1, 2], str(tmp_path / 'x.h5'))


@pytest.mark.skipif(not os.path.isfile('/root/reference/TranscranialModeling/MapPichardo.h5'), reason='reference tree not present')
def test_reads_the_reference_file_itself(golden):
    """Blosc-compressed, written by the real H5pySimple: only where the reference tree is mounted."""
    g, meta = golden
    mp = DF.ReadFromH5py('/root/reference/TranscranialModeling/MapPichardo.h5', use_h5py=False)
    assert sorted(mp) == sorted(meta['h5pysimple_layout'])
    assert np.array_equal(mp['rho'][:6], g['h5_pichardo_rho_head'])
    assert np.array_equal(mp['MapSoS'][:3, :3], g['h5_pichardo_sos_corner'])

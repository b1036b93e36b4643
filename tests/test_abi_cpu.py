"""The C-ABI library loads and exports every symbol include/babelfdtd.h declares; host-only entry
points (no GPU needed) agree with the oracle. No compute call is made here."""
import os
import re

import numpy as np
import pytest

from babelbrain_amd import _engine, harness as H
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    if not os.path.exists(_engine.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return _engine.load_library()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, 'include', 'babelfdtd.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = sorted(set(re.findall(r'\b(bfd_[a-z_0-9]+)\s*\(', hdr)))
    assert declared == sorted(_engine.ABI_SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s
    assert lib.bfd_abi_version() == 7


def test_struct_layout_matches_header():
    hdr = open(os.path.join(ROOT, 'include', 'babelfdtd.h')).read()
    body = hdr[hdr.index('typedef struct bfd_config {'):hdr.index('} bfd_config;')]
    body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
    names = []
    for decl in re.findall(r'(?:int32_t|uint32_t|double)\s+([^;]+);', body):
        names += [n.strip() for n in decl.split(',')]
    assert names == [f[0] for f in _engine.Config._fields_]


def test_stable_dt_and_tables_host_side(lib):
    ml = H.ct_material_rows(500e3, 32)
    h = H.spatial_step(500e3, 6)
    q = np.ones(len(ml)); q[2:] = 3
    for corr in (True, False):
        dt = _engine.stable_dt(ml, 500e3, corr, h, 0.5, q)
        assert dt == O.stable_dt(ml, 500e3, corr, h, 0.5, q)
        th, ch, cm = _engine.material_tables(ml, 500e3, corr, h, dt, q)
        to, co, cmo = O.tables(ml, 500e3, h, dt, corr, q)
        assert np.array_equal(th, to) and np.array_equal(ch, co) and cm == cmo
    # CFL: cmax dt / h = 0.5 * (6/7)/sqrt(3)
    assert abs(cm * dt / h - 0.5 * 6 / 7 / np.sqrt(3)) < 1e-12


def test_engine_refuses_without_gpu(lib):
    if lib.bfd_device_count() > 0:
        pytest.skip('a GPU is present')
    with pytest.raises(_engine.EngineError):
        _engine.Engine(64, 64, 64, 1, 1e-3, 1e-7, 5e5, 10)


def test_backend_entry_points_of_the_reference_exist(lib):
    """The names BabelBrain imports from the solver package's tools (BabelBrain.py:429-439, SelFiles.py:245-263)."""
    import babelbrain_amd
    from babelbrain_amd import RayleighAndBHTE as R
    for name in ('InitCuda', 'InitOpenCL', 'InitMetal', 'InitMLX', 'ForwardSimple', 'GenerateFocusTx', 'SpeedofSoundWater', 'BHTE',
                 'BHTEMultiplePressureFields'):
        assert callable(getattr(R, name)), name
    names = babelbrain_amd.ListDevices()
    assert isinstance(names, list) and len(names) == lib.bfd_device_count()
    if not names:
        with pytest.raises(_engine.EngineError):
            R.InitMLX('MI355X')

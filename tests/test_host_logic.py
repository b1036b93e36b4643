"""Host-side helpers of the drop-in (no GPU): slab partition, source compaction, sensor bookkeeping, BHTE schedule."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from babelbrain_amd import slab
from babelbrain_amd.PropagationModel import compact_sources, material_slab, n_steps, sensor_steps
from babelbrain_amd.RayleighAndBHTE import field_schedule


@given(st.integers(4, 4000), st.integers(1, 16))
@settings(max_examples=200, deadline=None)
def test_partition_is_a_balanced_cover(N3, world):
    if N3 < slab.MIN_PLANES * world:
        with pytest.raises(ValueError):
            slab.partition(N3, world)
        return
    parts = slab.partition(N3, world)
    assert parts[0][0] == 0 and sum(nk for _, nk in parts) == N3
    assert all(parts[r][0] + parts[r][1] == parts[r + 1][0] for r in range(world - 1))
    sizes = [nk for _, nk in parts]
    assert max(sizes) - min(sizes) <= 1 and min(sizes) >= slab.MIN_PLANES


@given(st.integers(0, 10 ** 6), st.integers(1, 12), st.integers(0, 200))
@settings(max_examples=200, deadline=None)
def test_sensor_steps_follow_the_sampling_rule(nt, sub, start):
    steps = sensor_steps(nt, sub, start)
    assert np.all(steps % sub == 0) and np.all(steps // sub >= start) and np.all(steps < max(nt, 0) + (nt == 0))
    assert len(steps) == max(0, (nt + sub - 1) // sub - start)


def test_n_steps_matches_the_callers_time_plan():
    dt = 1 / 500e3 / 35
    for nt in (1, 2, 70, 6755, 6760):
        assert n_steps(nt * dt, dt) == nt                         # TimeSimulation = nt*dt (BASE:2089)


def test_make_problem_keeps_the_callers_plan_beside_an_overridden_step_count():
    """bench.py's production_schedule weights its two rates by the caller's own time plan (BASE:2082-2109): info['plan_nt'] steps, of
    which the last 2 periods accumulate, whatever `steps` the bench asks for."""
    from tests.util import oracle_dt
    from babelbrain_amd import harness as H
    a, k, info = H.make_problem('C1', N=(48, 48, 64), steps=30, stable_dt_fn=oracle_dt)
    a2, k2, info2 = H.make_problem('C1', N=(48, 48, 64), stable_dt_fn=oracle_dt)
    assert info['nt'] == 30 and info2['nt'] == info2['plan_nt'] == info['plan_nt']
    assert info['plan_accumulating_steps'] == info2['plan_nt'] - k2['SensorStart'] * k2['SensorSubSampling']
    assert info['plan_accumulating_steps'] == 2 * info['ppp'] and info['plan_nt'] % info['ppp'] == 0


@given(st.integers(0, 2 ** 32 - 1))
@settings(max_examples=50, deadline=None)
def test_compact_sources_per_slab_union_is_the_whole(seed):
    rng = np.random.default_rng(seed)
    N = (int(rng.integers(3, 9)), int(rng.integers(3, 9)), int(rng.integers(8, 20)))
    smap = (rng.uniform(size=N) < 0.1) * rng.integers(1, 5, size=N)
    smap = smap.astype(np.uint32)
    Oz = rng.uniform(0.5, 1.5, N)
    whole = compact_sources(smap, np.array([0.0]), np.array([1]), Oz)
    lin, row, wx, wy, wz = whole
    assert np.all(np.diff(lin.astype(np.int64)) > 0)               # sorted by x-fastest voxel index
    i, j, k = lin % N[0], (lin // N[0]) % N[1], lin // (N[0] * N[1])
    assert np.array_equal(row, smap[i, j, k] - 1) and wy is None and np.all(wx == 0) and np.allclose(wz, Oz[i, j, k].astype(np.float32))
    seen = []
    for k0, nk in slab.partition(N[2], 2):
        l2, r2, _, _, _ = compact_sources(smap, np.array([0.0]), np.array([1]), Oz, k0, nk)
        assert np.all(l2 < N[0] * N[1] * nk)
        seen += list(l2.astype(np.int64) + k0 * N[0] * N[1])
    assert seen == list(lin.astype(np.int64))


def test_material_slab_ghost_planes():
    mm = np.arange(4 * 3 * 10).reshape(4, 3, 10)
    v, gl, gh = material_slab(mm, 0, 5)
    assert (gl, gh) == (0, 2) and v.shape[2] == 7
    v, gl, gh = material_slab(mm, 5, 5)
    assert (gl, gh) == (2, 0) and np.array_equal(v[:, :, 0], mm[:, :, 3])
    v, gl, gh = material_slab(mm, 1, 8)
    assert (gl, gh) == (1, 1) and v.shape[2] == 10


@given(st.lists(st.tuples(st.integers(0, 6), st.integers(0, 6)), min_size=1, max_size=5), st.integers(0, 80))
@settings(max_examples=200, deadline=None)
def test_field_schedule_properties(onoff, total):
    if sum(a + b for a, b in onoff) == 0:
        with pytest.raises(ValueError):
            field_schedule(onoff, total)
        return
    s = field_schedule(onoff, total)
    assert len(s) == total and set(np.unique(s)) <= set(range(-1, len(onoff)))
    period = sum(a + b for a, b in onoff)
    if total > period:
        assert np.array_equal(s[period:], s[:total - period])       # periodic
    first = s[:period]
    for n, (on, off) in enumerate(onoff):
        assert np.count_nonzero(first == n) == (on if total >= period else np.count_nonzero(first == n))


def test_bhte_source_forms():
    """The finite-voxel (exponential) heat source tends to the linear one for h*alpha -> 0 and is smaller by about
    h*alpha otherwise; everything else of the coefficient set is unchanged."""
    from babelbrain_amd import RayleighAndBHTE as R
    ml = {'Density': np.array([1000.0, 1896.5]), 'SoS': np.array([1500.0, 2476.0]), 'Attenuation': np.array([0.05, 81.0]),
          'SpecificHeat': np.array([4178.0, 1313.0]), 'Conductivity': np.array([0.6, 0.32]), 'Perfusion': np.array([0.0, 10.0]),
          'Absorption': np.array([0.85, 0.16])}
    dx, dt = 0.4e-3, 0.01
    cd, cp, q_lin = R.bhte_coefficients(ml, dx, dt, 0.5)
    cd2, cp2, q_exp = R.bhte_coefficients(ml, dx, dt, 0.5, source_form='exponential')
    assert np.array_equal(cd, cd2) and np.array_equal(cp, cp2)
    ratio = q_exp.astype(np.float64) / q_lin
    assert abs(ratio[0] - 1.0) < 1e-4                                     # water: h*alpha = 2e-5
    ha = dx * 81.0
    assert abs(ratio[1] - (1 - np.exp(-2 * ha)) / (2 * ha)) < 1e-6 and 0.95 < ratio[1] < 0.98
    with pytest.raises(ValueError):
        R.bhte_coefficients(ml, dx, dt, 0.5, source_form='other')


def test_thermal_host_logic():
    """Host side of the bio-heat drop-in (no device needed): the on/off schedule of steered multi-point sonications
    (CalculateTemperatureEffects.py:715-736), the per-material coefficients and their stability check."""
    import pytest
    from babelbrain_amd import RayleighAndBHTE as R
    assert R.field_schedule(np.array([[2, 1], [1, 2]]), 10).tolist() == [0, 0, -1, 1, -1, -1, 0, 0, -1, 1]
    assert R.field_schedule([[3, 0]], 4).tolist() == [0, 0, 0, 0]
    assert R.field_schedule([[1, 1]], 0).tolist() == []
    with pytest.raises(ValueError):
        R.field_schedule([[0, 0]], 4)
    with pytest.raises(ValueError):
        R.field_schedule([[2, -1]], 4)
    ml = {'Density': np.array([1000.0, 1041.0]), 'SoS': np.array([1500.0, 1562.0]), 'Attenuation': np.array([0.0, 3.45]),
          'SpecificHeat': np.array([4178.0, 3630.0]), 'Conductivity': np.array([0.6, 0.51]), 'Perfusion': np.array([0.0, 559.0]),
          'Absorption': np.array([0.0, 0.85]), 'InitTemperature': np.array([37.0, 37.0])}
    dx, dt = 5e-4, 0.05
    cd, cp, qf = R.bhte_coefficients(ml, dx, dt, 0.3)
    assert cd.dtype == cp.dtype == qf.dtype == np.float32
    assert abs(cd[1] / (dt * 0.51 / (1041.0 * 3630.0 * dx ** 2)) - 1) < 1e-6
    assert cp[0] == 0 and abs(cp[1] / (dt * 1050.0 * 3617.0 * 559.0 / (6e7 * 3630.0)) - 1) < 1e-6
    assert qf[0] == 0 and abs(qf[1] / (dt * 0.3 * 0.85 * 3.45 / (1041.0 * 1562.0) / (1041.0 * 3630.0)) - 1) < 1e-6
    # the exponential source form tends to the linear one for thin voxels and stays below it
    _, _, qe = R.bhte_coefficients(ml, dx, dt, 0.3, source_form='exponential')
    assert 0.99 < qe[1] / qf[1] < 1.0
    with pytest.raises(ValueError):
        R.bhte_coefficients(ml, dx, dt, 0.3, source_form='other')
    with pytest.raises(ValueError):
        R.bhte_coefficients(ml, 1e-4, 1.0)                 # dt k/(rho c dx^2) > 1/6: the explicit scheme would blow up
    # device list of the Rayleigh integral: parsing only (no device is touched until a call is made)
    R.set_devices('0, 0,0')
    assert R._devices == [0, 0, 0]
    R.set_devices([3])
    assert R._devices is None and R._device == 3
    R.set_devices(None)
    R._device = 0
    assert R._devices is None


def test_benchmark_test_file_hook(tmp_path):
    """The reference's synthetic-medium hook (BenchmarkTestFile, BASE:1253-1260, 1313-1321): a medium written in that layout and
    read back gives make_problem the very inputs the in-code medium gives it."""
    from babelbrain_amd import harness as H
    from oracle import oracle as O
    dt_fn = lambda ml, f, h, c: O.stable_dt(ml, f, True, h, c)
    a, k, info = H.make_problem('C2', N=(48, 40, 56), steps=40, stable_dt_fn=dt_fn)
    path = str(tmp_path / 'bench_medium.h5')
    H.save_benchmark_medium(path, a[0], a[1], k['QCorrection'], TestType=1)
    mm, ml, q, sos = H.benchmark_medium(path)
    assert np.array_equal(mm, a[0]) and np.array_equal(ml, a[1]) and np.array_equal(q, np.asarray(k['QCorrection'], np.float64))
    assert sos == min(ml[:, 1].min(), ml[ml[:, 2] > 0, 2].min())
    a2, k2, info2 = H.make_problem('C2', steps=40, stable_dt_fn=dt_fn, benchmark_file=path)
    assert info2['N'] == (48, 40, 56) and info2['medium'] == 'benchmark file'
    for x, y in zip(a, a2):
        assert np.array_equal(np.asarray(x), np.asarray(y))
    assert k2['DT'] == k['DT'] and np.array_equal(k2['QCorrection'], k['QCorrection'])
    with pytest.raises(ValueError):
        H.save_benchmark_medium(path, np.zeros((4, 4, 4), np.uint32), a[1])      # a material that never occurs in the map


def test_peer_status_words_decode():
    """bfd_group_peer_status packs, per interface, the path (low 4 bits: 0 same device, 1 direct, 2 staged through the host) and
    can-access / enabled bits for each direction; the decoder must flag a staged interface and keep the order of the interfaces."""
    from babelbrain_amd._engine import decode_peer_status
    words = [0, 1 | 16 | 32 | 64 | 128, 2 | 16 | 32, 2]
    d = decode_peer_status(words, devices=[0, 0, 1, 2, 3])
    assert [e['interface'] for e in d] == [0, 1, 2, 3]
    assert d[0]['path'].startswith('same device') and d[0]['direct'] and d[0]['devices'] == [0, 0]
    assert d[1]['direct'] and d[1]['can_access'] == [True, True] and d[1]['enabled'] == [True, True] and d[1]['devices'] == [0, 1]
    assert not d[2]['direct'] and d[2]['path'].startswith('STAGED') and d[2]['can_access'] == [True, False] and d[2]['enabled'] == [True, False]
    assert not d[3]['direct'] and d[3]['can_access'] == [False, False]
    assert decode_peer_status([]) == []


def test_bench_strong_c5_block_anchors_itself(monkeypatch):
    """bench.py's `strong_c5` block (the 1024^3 curve north_star names) carries its own one-device anchor and scaling_efficiency =
    value / (distinct devices x anchor); at N = 1 the block is the anchor; a failing anchor or headline still yields a block."""
    import argparse
    import bench
    calls = []

    def fake_group_run(args, config, N, ndev, dt_fn, steps, warmup, windows, variant, label):
        calls.append((config, tuple(N), ndev))
        if ndev == 1 and fail.get('anchor'):
            raise RuntimeError('no memory')
        if ndev > 1 and fail.get('head'):
            raise RuntimeError('peer access')
        return {'value': 70000.0 * ndev * (0.9 if ndev > 1 else 1.0), 'ms_per_step': 15.0 / ndev, 'distinct_devices': ndev, 'emulated': False,
                'array_placement_slab0': 'note'}
    monkeypatch.setattr(bench, 'group_run', fake_group_run)
    args = argparse.Namespace(strong_c5_steps=30, warmup=50)
    fail = {}
    b1 = bench.strong_c5(args, 1, None, 0)
    assert calls == [('C5', (1024, 1024, 1024), 1)] and b1['scaling_efficiency'] == 1.0 and b1['one_device_same_volume']['value'] == b1['value']
    calls.clear()
    b4 = bench.strong_c5(args, 4, None, 0)
    assert [c[2] for c in calls] == [1, 4] and b4['scaling'] == 'strong'
    assert abs(b4['scaling_efficiency'] - 0.9) < 1e-12 and b4['one_device_same_volume']['value'] == 70000.0
    fail = {'anchor': True}
    b = bench.strong_c5(args, 4, None, 0)
    assert b['value'] > 0 and b['scaling_efficiency'] is None and 'no memory' in b['one_device_same_volume']['error']
    fail = {'head': True}
    b = bench.strong_c5(args, 4, None, 0)
    assert b['value'] is None and 'peer access' in b['error'] and b['one_device_same_volume']['value'] == 70000.0


def test_bench_stdout_carries_only_the_line(tmp_path):
    """Libraries write to file descriptor 1 (gloo announces its connections there): after claim_stdout() such output lands on stderr and the
    JSON line is the only thing on the process's stdout."""
    import subprocess, sys, os, json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys; sys.path.insert(0, %r); import bench; bench.claim_stdout(); os.write(1, b'[Gloo] Rank 0 is connected\\n'); "
            "print('chatter'); bench.emit_line({'metric': 'm', 'value': 1.0})" % root)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout) == {'metric': 'm', 'value': 1.0}
    assert '[Gloo] Rank 0 is connected' in r.stderr and 'chatter' in r.stderr


def test_bench_line_fits_the_drivers_record_and_roofline_carries_the_rounds_numbers():
    """The driver's record keeps `roofline`, `config` and `cpu_baseline` of the line whole and only a 2000-byte tail of the rest: the numbers a
    reader needs to recompute the shear step and the 1024^3 anchor are flat scalars inside `roofline` (bench.roofline_summary), and the line
    on stdout stays under 8 KB (bench.compact_line; the full text goes to stderr) without losing a headline number."""
    import json, os
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    full = json.load(open(os.path.join(root, 'profiles', 'r5', 'bench_default_n1.json')))      # a real line of this bench (13 KB)
    line = json.loads(json.dumps(full))
    bench.roofline_summary(line)
    r = line['roofline']
    sh = full['shear_workload']
    assert r['shear512_value'] == sh['value'] and r['shear512_step_frac'] == sh['roofline_step']['frac']
    for c, row in sh['roofline_kernels'].items():
        assert r['shear512_%s_frac' % c] == row['frac'] and r['shear512_%s_ms' % c] == row['avg_launch_ms']
        assert abs(r['shear512_%s_moved_over_alg' % c] - row['traffic_from_profile'] / row['algorithmic_bytes_per_launch']) < 1e-12
    assert r['step_frac'] == full['roofline_step']['frac'] and r['c5_value'] == full['strong_c5']['value'] and r['c5_one_device_value'] == r['c5_value']
    assert all(not isinstance(v, (dict, list)) for k, v in r.items() if k != 'profile_ref')
    short = bench.compact_line(line)
    assert len(json.dumps(line)) > 8192 > len(json.dumps(short))
    assert set(short) == set(line)
    for k in ('metric', 'unit', 'n_gpus', 'steps', 'warmup', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data'):
        assert short[k] == line[k]
    assert abs(short['value'] / line['value'] - 1) < 1e-5 and abs(short['roofline']['frac'] / line['roofline']['frac'] - 1) < 1e-5
    assert short['config']['workload'] == line['config']['workload'] and short['cpu_baseline']['kind'] == 'port'
    assert abs(short['roofline']['shear512_velocity_solid_frac'] / r['shear512_velocity_solid_frac'] - 1) < 1e-5
    # a line already short is not touched
    assert bench.compact_line({'metric': 'm', 'value': 1.0}) == {'metric': 'm', 'value': 1.0}


def test_bench_watchdog_prints_the_line_it_has():
    """bench.py --watchdog-seconds: a block that hangs after the headline was measured must not cost the line (rank 0 prints what it has and
    ends the process); without a headline the line says so and the exit code is not 0."""
    import json, subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    head = "import bench, time, argparse; a = argparse.Namespace(watchdog_seconds=0.3, steps=20, warmup=5, scaling='weak', config='C3'); bench.start_watchdog(a, 1); "
    r = subprocess.run([sys.executable, '-c', head + "bench.watch_line({'metric': 'm', 'value': 5.0, 'n_gpus': 1}); time.sleep(20); print('not reached')"],
                       cwd=root, capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and 'not reached' not in r.stdout
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line['value'] == 5.0 and 'expired' in line['watchdog']
    r = subprocess.run([sys.executable, '-c', head + "time.sleep(20); print('not reached')"], cwd=root, capture_output=True, text=True, timeout=60)
    assert r.returncode == 3 and 'not reached' not in r.stdout
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line['value'] is None and line['n_gpus'] == 1 and 'no headline' in line['error']
    # a run that finishes first prints exactly one line
    r = subprocess.run([sys.executable, '-c', head + "bench.watch_line({'value': 1.0}); ok = bench.stop_watchdog(); time.sleep(0.6); print('finished', ok)"],
                       cwd=root, capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.strip().splitlines() == ['finished True']

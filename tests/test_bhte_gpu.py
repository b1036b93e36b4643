"""Device bio-heat solver (bfd_bhte_run through babelbrain_amd.RayleighAndBHTE.BHTE) against the numpy oracle and
analytic answers. Tolerance 1e-5 relative L2 on the temperature rise and the dose (float32 on both sides, same
operation order; powf differs in the last bits)."""
import numpy as np
import pytest

from oracle import bhte_oracle as BO
from tests.util import rel_l2

pytestmark = pytest.mark.gpu


def _ftz(a):
    """The library's float32 arithmetic flushes subnormal results to zero (csrc/Makefile); numpy keeps them. A heat increment
    below 1.2e-38 K per step is zero either way for the temperature."""
    a = np.array(a, np.float32)
    a[np.abs(a) < np.finfo(np.float32).tiny] = 0
    return a


def _materials():
    # water, skin, cortical, trabecular, brain rows of CalculateTemperatureEffects.py:780-791; acoustic columns of MatFreq[500e3]
    return {'Density': np.array([1000.0, 1116.0, 1896.5, 1738.0, 1041.0]), 'SoS': np.array([1500.0, 1537.0, 2476.0, 2205.0, 1562.0]),
            'Attenuation': np.array([0.0, 2.3, 81.0, 81.0, 3.45]), 'SpecificHeat': np.array([4178.0, 3391.0, 1313.0, 2274.0, 3630.0]),
            'Conductivity': np.array([0.6, 0.37, 0.32, 0.31, 0.51]), 'Perfusion': np.array([0.0, 106.0, 10.0, 30.0, 559.0]),
            'Absorption': np.array([0.0, 0.85, 0.16, 0.15, 0.85]), 'InitTemperature': np.full(5, 37.0)}


def test_bhte_matches_oracle_and_monitors():
    from babelbrain_amd import RayleighAndBHTE as R
    rng = np.random.default_rng(5)
    N = (40, 36, 44)
    ml = _materials()
    mm = rng.integers(0, 5, N).astype(np.uint8)
    x, y, z = np.meshgrid(*[np.arange(n) - n / 2 for n in N], indexing='ij')
    p = 5.0e6 * np.exp(-(x ** 2 + y ** 2 + (z / 2) ** 2) / 30.0)
    mpm = np.zeros(N, np.uint32); mpm[20, 18, 22] = 1; mpm[10, 10, 30] = 2; mpm[25, 20, 12] = 3
    dx, dt, nS, nOn = 4e-4, 0.02, 120, 80
    T, D, mon, Q, pts = R.BHTE(p, mm, ml, dx, nS, nOn, 18, nFactorMonitoring=10, dt=dt, DutyCycle=0.5, MonitoringPointsMap=mpm)
    cd, cp, qf = R.bhte_coefficients(ml, dx, dt, 0.5)
    q = (p.astype(np.float32) ** 2) * qf[mm]
    T0 = np.full(N, 37.0, np.float32)
    To, Do = BO.bhte(T0, np.zeros(N, np.float32), q, mm, cd, cp, 37.0, dt, nS, nOn)
    assert To.max() > 40.0
    assert rel_l2(T - 37.0, To - 37.0) < 1e-5 and rel_l2(D, Do) < 1e-5
    assert np.array_equal(Q, _ftz(q))
    assert mon.shape == (N[0], N[2], 12) and pts.shape == (3, nS)
    # the last monitored sample is step 110: compare with a shorter oracle run
    T110, _ = BO.bhte(T0, np.zeros(N, np.float32), q, mm, cd, cp, 37.0, dt, 111, nOn)
    assert rel_l2(mon[:, :, 11] - 37.0, T110[:, 18, :] - 37.0) < 1e-5
    assert abs(pts[0, -1] - T[20, 18, 22]) < 1e-6 and abs(pts[1, -1] - T[10, 10, 30]) < 1e-6


def test_bhte_analytic_limits():
    from babelbrain_amd import RayleighAndBHTE as R
    N = (24, 24, 24)
    ml = _materials()
    one = {k: v[4:5].copy() for k, v in ml.items()}                 # brain only
    mm = np.zeros(N, np.uint8)
    dx, dt = 1e-3, 0.05
    # (i) no conduction, no perfusion, uniform pressure: linear rise dT = n dt a_abs p^2/(rho c_s)/(rho c)
    a = dict(one); a['Conductivity'] = np.array([0.0]); a['Perfusion'] = np.array([0.0])
    p = np.full(N, 5e5)
    T, D, _, Q = R.BHTE(p, mm, a, dx, 200, 200, -1, dt=dt)
    rate = 0.85 * 3.45 * (5e5) ** 2 / (1041.0 * 1562.0) / (1041.0 * 3630.0)
    assert abs((T[12, 12, 12] - 37.0) / (200 * dt * rate) - 1) < 1e-4
    # (ii) perfusion only from 40 degC: exponential return to the core temperature
    b = dict(one); b['Conductivity'] = np.array([0.0]); b['InitTemperature'] = np.array([40.0])
    T, D, _, Q = R.BHTE(p * 0, mm, b, dx, 400, 0, -1, dt=dt, stableTemp=37.0)
    w = 1050.0 * 3617.0 * 559.0 / 6e7 / 3630.0
    assert abs((T[12, 12, 12] - 37.0) / (3.0 * np.exp(-w * 400 * dt)) - 1) < 2e-3
    # (iii) dose at constant 45 degC: dt/60 * 0.5^(43-45) per step
    c = dict(a); c['InitTemperature'] = np.array([45.0])
    T, D, _, Q = R.BHTE(p * 0, mm, c, dx, 120, 0, -1, dt=dt, stableTemp=45.0)
    assert abs(D[12, 12, 12] / (120 * dt / 60 * 4.0) - 1) < 1e-5
    # (iv) conduction conserves heat away from the faces and spreads a hot spot symmetrically
    d = dict(one); d['Perfusion'] = np.array([0.0])
    T0 = np.full(N, 37.0, np.float32); T0[12, 12, 12] = 47.0
    T, D, _, Q = R.BHTE(p * 0, mm, d, dx, 60, 0, -1, dt=dt, initT0=T0)
    assert abs((T.astype(np.float64) - 37.0).sum() / 10.0 - 1) < 1e-3 and T[12, 12, 12] < 47.0      # float32 cells near 37
    assert abs(T[11, 12, 12] - T[13, 12, 12]) < 1e-6 and abs(T[12, 11, 12] - T[12, 12, 13]) < 1e-6
    with pytest.raises(ValueError):
        R.BHTE(p, mm, d, 1e-4, 10, 0, -1, dt=1.0)


def test_multiple_pressure_fields_schedule_and_oracle():
    """BHTEMultiplePressureFields (CalculateTemperatureEffects.py:381-394, 978-990): three steered focal spots
    heating in turn, schedule built like the caller does (equal on/off slots, :715-736)."""
    from babelbrain_amd import RayleighAndBHTE as R
    # the schedule itself (host logic)
    s = R.field_schedule(np.array([[2, 1], [1, 2]]), 10)
    assert s.tolist() == [0, 0, -1, 1, -1, -1, 0, 0, -1, 1]
    assert R.field_schedule([[3, 0]], 4).tolist() == [0, 0, 0, 0]
    with pytest.raises(ValueError):
        R.field_schedule([[0, 0]], 4)

    rng = np.random.default_rng(9)
    N = (36, 40, 44)
    ml = _materials()
    mm = rng.integers(0, 5, N).astype(np.uint8)
    x, y, z = np.meshgrid(*[np.arange(n) - n / 2 for n in N], indexing='ij')
    fields = np.stack([4.0e6 * np.exp(-((x - cx) ** 2 + (y - cy) ** 2 + (z / 2) ** 2) / 25.0) for cx, cy in ((-6, 0), (5, 4), (0, -7))])
    onoff = np.array([[7, 5], [7, 5], [7, 5]], np.int32)
    dx, dt, nS = 4e-4, 0.02, 130
    mpm = np.zeros(N, np.uint32); mpm[12, 20, 22] = 1; mpm[23, 24, 22] = 2
    T, D, mon, Q, pts = R.BHTEMultiplePressureFields(fields, mm, ml, dx, nS, onoff, 20, nFactorMonitoring=13, dt=dt, MonitoringPointsMap=mpm)
    cd, cp, qf = R.bhte_coefficients(ml, dx, dt, 1.0)
    q = np.stack([(f.astype(np.float32) ** 2) * qf[mm] for f in fields])
    sched = R.field_schedule(onoff, nS)
    T0 = np.full(N, 37.0, np.float32)
    To, Do = BO.bhte(T0, np.zeros(N, np.float32), q, mm, cd, cp, 37.0, dt, nS, 0, field_of_step=sched)
    assert To.max() > 39.0
    assert rel_l2(T - 37.0, To - 37.0) < 1e-5 and rel_l2(D, Do) < 1e-5
    assert Q.shape == fields.shape and np.array_equal(Q, _ftz(q))
    assert mon.shape == (N[0], N[2], 10) and pts.shape == (2, nS)
    # each monitored point heats fastest while its own focal spot is on
    rise = np.diff(np.concatenate([[37.0], pts[0]]))
    assert rise[sched == 0].mean() > 3 * rise[sched == 1].mean()
    # one field through the multi-field entry point == the single-field call with the same on-steps and duty cycle 1
    T1, D1, _, Q1 = R.BHTEMultiplePressureFields(fields[:1], mm, ml, dx, 60, [[40, 20]], -1, dt=dt)
    T2, D2, _, Q2 = R.BHTE(fields[0], mm, ml, dx, 60, 40, -1, dt=dt, DutyCycle=1.0)
    assert np.array_equal(T1, T2) and np.array_equal(D1, D2) and np.array_equal(Q1[0], Q2)
    # continuation from a previous group (initT0/initDose) is the same as one longer run (schedule phase restarts)
    Ta, Da, _, _ = R.BHTEMultiplePressureFields(fields, mm, ml, dx, 36, onoff, -1, dt=dt)
    Tb, Db, _, _ = R.BHTEMultiplePressureFields(fields, mm, ml, dx, 36, onoff, -1, dt=dt, initT0=Ta, initDose=Da)
    Tc, Dc, _, _ = R.BHTEMultiplePressureFields(fields, mm, ml, dx, 72, onoff, -1, dt=dt)
    assert np.array_equal(Tb, Tc) and np.array_equal(Db, Dc)
    with pytest.raises(ValueError):
        R.BHTEMultiplePressureFields(fields, mm, ml, dx, 10, [[1, 1]], -1, dt=dt)


def test_two_steps_per_launch_equal_one_step_per_launch(monkeypatch):
    """The default path takes two steps per launch (bhte_step2g: z-marching tiles of 64 x 24 cells with two rings, loads in
    flight across the plane; BFD_BHTE_KERNEL=1: the round-3 kernel bhte_step2 on 64 x 26 tiles); the temperature, the dose and
    every monitor must have the bits of the one-step kernel, on grids that do not fill the tiles, with odd step counts, with
    the field changing between the two fused steps, and with monitors on intermediate steps."""
    from babelbrain_amd import RayleighAndBHTE as R
    rng = np.random.default_rng(11)
    ml = _materials()
    for N, nS, onoff, zrun in (((150, 61, 37), 25, [[3, 2], [2, 1]], None), ((64, 26, 16), 8, [[1, 0], [1, 1]], None),
                               ((70, 30, 35), 13, [[5, 3]], '5'), ((3, 3, 3), 5, [[2, 1]], None), ((131, 55, 20), 7, [[1, 1], [1, 0], [2, 2]], '32')):
        mm = rng.integers(0, 5, N).astype(np.uint8)
        fields = (3.0e6 * rng.random((len(onoff),) + N)).astype(np.float32)
        mpm = np.zeros(N, np.uint32); mpm[1, 1, 1] = 1; mpm[N[0] // 2, N[1] // 2, N[2] // 2] = 2; mpm[N[0] - 1, N[1] - 2, 0] = 3
        T0 = (37.0 + 8.0 * rng.random(N)).astype(np.float32)                 # some cells above 43: both dose bases
        out = {}
        for fuse, kernel in (('1', '0'), ('1', '1'), ('0', '0')):
            monkeypatch.setenv('BFD_BHTE_FUSE', fuse)
            monkeypatch.setenv('BFD_BHTE_KERNEL', kernel)
            if zrun: monkeypatch.setenv('BFD_BHTE_ZRUN', zrun)
            else: monkeypatch.delenv('BFD_BHTE_ZRUN', raising=False)
            out[fuse + kernel] = R.BHTEMultiplePressureFields(fields, mm, ml, 4e-4, nS, onoff, N[1] // 2, nFactorMonitoring=3, dt=0.02, initT0=T0, MonitoringPointsMap=mpm)
        for k in ('10', '11'):
            for a, b in zip(out[k], out['00']):
                assert np.array_equal(a, b)
        assert out['10'][0].max() > 44.0 and out['10'][1].max() > 0
    monkeypatch.delenv('BFD_BHTE_KERNEL', raising=False)
    # and the oracle, bit for bit on the temperature now that the roundings are pinned (the dose goes through exp2f against numpy's power)
    N = (70, 30, 35)
    mm = rng.integers(0, 5, N).astype(np.uint8)
    p = (3.0e6 * rng.random(N)).astype(np.float32)
    monkeypatch.setenv('BFD_BHTE_FUSE', '1')
    T, D, _, Q = R.BHTE(p, mm, ml, 4e-4, 11, 6, -1, dt=0.02)
    cd, cp, qf = R.bhte_coefficients(ml, 4e-4, 0.02, 1.0)
    To, Do = BO.bhte(np.full(N, 37.0, np.float32), np.zeros(N, np.float32), Q, mm, cd, cp, 37.0, 0.02, 11, 6)
    assert np.array_equal(T, To) and rel_l2(D, Do) < 1e-6


@pytest.mark.parametrize('steps', ['3', '4', 'default'])
def test_three_and_four_steps_per_pass_equal_one_step_per_launch(steps, monkeypatch):
    """Round 6: bhte_stepNg takes S = 4 (default; two cells per thread while a field heats, four while nothing does) or 3 steps per pass over regions of (64 + 2 S) x 28 cells wherever the next S steps carry the same heat
    field (or none) and no sample of the monitored plane falls strictly inside; the monitor points of the steps inside a pass come from cone_points
    (the cube around the point advanced level by level). Temperature, dose, monitored plane, heat source and point series must have the bits of one
    step per launch: long and short on / off stretches, several fields, plane samples every 1 .. 10 steps, points on faces and edges, grids at the
    tile edges (y tiles of 22 / 20 rows), run lengths from 1 plane up, and against the oracle."""
    from babelbrain_amd import RayleighAndBHTE as R
    rng = np.random.default_rng(23)
    ml = _materials()
    monkeypatch.delenv('BFD_BHTE_KERNEL', raising=False)
    cases = (((150, 61, 37), 40, [[9, 7], [5, 3]], None, 3, 30), ((64, 22, 16), 16, [[4, 4]], None, 1, 10), ((65, 23, 9), 13, [[7, 6]], '1', 4, 11),
             ((70, 41, 35), 27, [[12, 6], [3, 0]], '5', 10, -1), ((3, 3, 3), 9, [[4, 2]], None, 2, 1), ((131, 45, 20), 30, [[6, 4], [0, 5], [8, 2]], '32', 5, 22),
             ((129, 20, 50), 25, [[13, 12]], '64', 7, 0))
    for N, nS, onoff, zrun, fm, sl in cases:
        mm = rng.integers(0, 5, N).astype(np.uint8)
        fields = (3.0e6 * rng.random((len(onoff),) + N)).astype(np.float32)
        mpm = np.zeros(N, np.uint32); mpm[1, 1, 1] = 1; mpm[N[0] // 2, N[1] // 2, N[2] // 2] = 2; mpm[N[0] - 1, N[1] - 2, 0] = 3; mpm[0, N[1] - 1, N[2] - 1] = 4
        T0 = (37.0 + 8.0 * rng.random(N)).astype(np.float32)
        out = {}
        for name, env in (('S', dict(BFD_BHTE_STEPS=steps) if steps != 'default' else {}), ('one', dict(BFD_BHTE_FUSE='0'))):
            for k in ('BFD_BHTE_FUSE', 'BFD_BHTE_STEPS'): monkeypatch.delenv(k, raising=False)
            for k, v in env.items(): monkeypatch.setenv(k, v)
            if zrun: monkeypatch.setenv('BFD_BHTE_ZRUN', zrun)
            else: monkeypatch.delenv('BFD_BHTE_ZRUN', raising=False)
            out[name] = R.BHTEMultiplePressureFields(fields, mm, ml, 4e-4, nS, onoff, sl, nFactorMonitoring=fm, dt=0.02, initT0=T0, MonitoringPointsMap=mpm)
        for q, (a, b) in enumerate(zip(out['S'], out['one'])):
            assert np.array_equal(a, b), (N, nS, onoff, zrun, fm, q)
        assert out['S'][0].max() > 44.0 and out['S'][1].max() > 0 and out['S'][4].shape[1] == nS and out['S'][4].shape[0] in (3, 4)
    for k in ('BFD_BHTE_FUSE', 'BFD_BHTE_ZRUN'): monkeypatch.delenv(k, raising=False)
    if steps != 'default':
        monkeypatch.setenv('BFD_BHTE_STEPS', steps)
    N = (70, 30, 35)
    mm = rng.integers(0, 5, N).astype(np.uint8)
    p = (3.0e6 * rng.random(N)).astype(np.float32)
    T, D, _, Q = R.BHTE(p, mm, ml, 4e-4, 23, 13, -1, dt=0.02)
    cd, cp, qf = R.bhte_coefficients(ml, 4e-4, 0.02, 1.0)
    To, Do = BO.bhte(np.full(N, 37.0, np.float32), np.zeros(N, np.float32), Q, mm, cd, cp, 37.0, 0.02, 23, 13)
    assert np.array_equal(T, To) and rel_l2(D, Do) < 1e-6


def test_x_fastest_entry_points_of_the_c_abi():
    """bfd_bhte_run / bfd_bhte_run_fields keep their x-fastest contract (volumes [k][j][i], heat source precomputed); the
    drop-in goes through bfd_bhte_run_volumes (caller's C order). Same run both ways: equal up to the order in which the six
    neighbours are summed (fastest axis first there, axis 0 first here)."""
    import ctypes as C
    from babelbrain_amd import RayleighAndBHTE as R, _engine
    lib = _engine.load_library()
    rng = np.random.default_rng(21)
    N = (45, 30, 38)
    ml = _materials()
    mm = rng.integers(0, 5, N).astype(np.uint8)
    p = (3.0e6 * rng.random(N)).astype(np.float32)
    nS, nOn, dx, dt = 15, 9, 4e-4, 0.02
    mpm = np.zeros(N, np.uint32); mpm[7, 8, 9] = 1
    T, D, mon, Q, pts = R.BHTE(p, mm, ml, dx, nS, nOn, 11, nFactorMonitoring=4, dt=dt, MonitoringPointsMap=mpm)
    cd, cp, qf = R.bhte_coefficients(ml, dx, dt, 1.0)
    xf = lambda a, t: np.ascontiguousarray(a.transpose(2, 1, 0), dtype=t)
    mat, q = xf(mm, np.uint8), xf(Q, np.float32)
    Tx = np.full(mat.shape, 37.0, np.float32); Dx = np.zeros(mat.shape, np.float32)
    monx = np.zeros((N[0], N[2], 4), np.float32)
    idx = np.array([7 + N[0] * (8 + N[1] * 9)], np.uint32); ptx = np.zeros((1, nS), np.float32)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    ms = C.c_double()
    rc = lib.bfd_bhte_run(0, N[0], N[1], N[2], 5, ptr(mat), ptr(cd), ptr(cp), ptr(q), ptr(Tx), ptr(Dx), 37.0, dt, nS, nOn, 11, 4, ptr(monx),
                          1, ptr(idx), ptr(ptx), C.byref(ms))
    assert rc == 0, lib.bfd_last_error()
    assert rel_l2(Tx.transpose(2, 1, 0) - 37.0, T - 37.0) < 1e-6 and rel_l2(Dx.transpose(2, 1, 0), D) < 1e-6
    assert rel_l2(monx - 37.0, mon - 37.0) < 1e-6 and rel_l2(ptx - 37.0, pts - 37.0) < 1e-6
    # bad arguments are refused, not run
    assert lib.bfd_bhte_run_volumes(0, 2, 30, 38, 5, ptr(mat), ptr(cd), ptr(cp), ptr(qf), ptr(cd), 1, ptr(q), None, ptr(Tx), ptr(Dx), 0, 37.0, dt, 1,
                                    ptr(np.zeros(1, np.int32)), -1, 1, None, 0, None, None, None) == -1
    with pytest.raises(ValueError):
        R.BHTE(p, mm, ml, dx, nS, nOn, 30, dt=dt)                    # monitored plane outside the volume


def test_input_forms_the_caller_may_hand_over():
    """Fortran-ordered, float64 and uint32 inputs (what numpy slicing / transposes in a caller produce) give the result of the
    canonical float32 / uint8 C-ordered ones; volumes thinner than 3 cells and a monitoring map of another shape are refused."""
    from babelbrain_amd import RayleighAndBHTE as R, _engine
    rng = np.random.default_rng(31)
    N = (33, 47, 29)
    ml = _materials()
    mm = rng.integers(0, 5, N).astype(np.uint8)
    p = (2.5e6 * rng.random(N)).astype(np.float32)
    T0 = (37.0 + 2.0 * rng.random(N)).astype(np.float32)
    D0 = rng.random(N).astype(np.float32)
    mpm = np.zeros(N, np.uint32); mpm[5, 6, 7] = 2; mpm[20, 40, 3] = 1
    ref = R.BHTE(p, mm, ml, 4e-4, 9, 5, 10, nFactorMonitoring=2, dt=0.02, initT0=T0, initDose=D0, MonitoringPointsMap=mpm)
    alt = R.BHTE(np.asfortranarray(p.astype(np.float64)), np.asfortranarray(mm.astype(np.uint32)), ml, 4e-4, 9, 5, 10, nFactorMonitoring=2, dt=0.02,
                 initT0=np.asfortranarray(T0.astype(np.float64)), initDose=np.asfortranarray(D0), MonitoringPointsMap=np.asfortranarray(mpm.astype(np.int64)))
    for a, b in zip(ref, alt):
        assert a.shape == b.shape and np.array_equal(a, b)
    assert ref[4].shape == (2, 9) and ref[4][0, -1] == ref[0][20, 40, 3] and ref[4][1, -1] == ref[0][5, 6, 7]     # rows ordered by point id
    T0c, D0c = T0.copy(), D0.copy()
    R.BHTE(p, mm, ml, 4e-4, 4, 2, -1, dt=0.02, initT0=T0, initDose=D0)
    assert np.array_equal(T0, T0c) and np.array_equal(D0, D0c)
    with pytest.raises(_engine.EngineError):
        R.BHTE(p[:2], mm[:2], ml, 4e-4, 4, 2, -1, dt=0.02)
    with pytest.raises(ValueError):
        R.BHTE(p, mm, ml, 4e-4, 4, 2, -1, dt=0.02, MonitoringPointsMap=mpm[:-1])
    with pytest.raises(ValueError):
        R.BHTE(p, mm, ml, 4e-4, 4, 2, -1, dt=0.02, initT0=T0[:-1])

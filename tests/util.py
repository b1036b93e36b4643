import numpy as np

from oracle import oracle as O


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    den = np.sqrt(np.sum(b * b))
    if den == 0:
        return float(np.sqrt(np.sum(a * a)))
    return float(np.sqrt(np.sum((a - b) ** 2)) / den)


def oracle_dt(ml, f, h, acfl):
    return O.stable_dt(ml, f, True, h, acfl)


ALL_MAPS = ['Vx', 'Vy', 'Vz', 'Sigmaxx', 'Sigmayy', 'Sigmazz', 'Sigmaxy', 'Sigmaxz', 'Sigmayz', 'Pressure']


def compare_runs(out_hip, out_ref, tol=1e-5, both=False):
    """out_* are the tuples StaggeredFDTD_3D_with_relaxation returns. Returns the worst rel-L2."""
    worst = 0.0
    Sh, Lh = out_hip[0], out_hip[1]
    Sr, Lr = out_ref[0], out_ref[1]
    assert np.array_equal(out_hip[-1]['IndexSensorMap'], out_ref[-1]['IndexSensorMap'])
    np.testing.assert_allclose(Sh['time'], Sr['time'], rtol=0, atol=1e-15)
    dicts = [(Sh, Sr, 'sensor'), (Lh, Lr, 'last'), (out_hip[2], out_ref[2], 'rms/peak')]
    if both:
        dicts.append((out_hip[3], out_ref[3], 'peak'))
    for dh, dr, what in dicts:
        assert set(dh.keys()) == set(dr.keys()), (what, dh.keys(), dr.keys())
        for k in dr:
            if k == 'time':
                continue
            assert dh[k].shape == dr[k].shape, (what, k, dh[k].shape, dr[k].shape)
            assert dh[k].dtype == np.float32
            e = rel_l2(dh[k], dr[k])
            assert e <= tol, '%s[%s]: rel L2 %.3e > %.1e' % (what, k, e, tol)
            worst = max(worst, e)
    return worst

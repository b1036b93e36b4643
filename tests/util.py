import numpy as np

from oracle import oracle as O


def rel_l2(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    den = np.sqrt(np.sum(b * b))
    if den == 0:
        return float(np.sqrt(np.sum(a * a)))
    return float(np.sqrt(np.sum((a - b) ** 2)) / den)


def free_port():
    """A TCP port nobody listens on (rendezvous of spawned ranks on 127.0.0.1)."""
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def run_ranks(worker, world, timeout=800, extra=()):
    """Spawn `world` processes running worker(rank, world, port, queue, *extra), return what rank 0 put on the queue.
    Polls the children while waiting (a crashed rank fails the test at once instead of stalling it) and never leaves
    orphans behind."""
    import queue
    import time
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, q) + tuple(extra)) for r in range(world)]
    for p in procs:
        p.start()
    result = None
    try:
        deadline = time.time() + timeout
        while result is None:
            try:
                result = q.get(timeout=2)
            except queue.Empty:
                dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
                assert not dead, 'a rank died (exit codes %s)' % dead
                assert time.time() < deadline, 'ranks timed out'
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
    finally:
        for p in procs:
            if p.is_alive():
                p.terminate()
            p.join(10)
    return result


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

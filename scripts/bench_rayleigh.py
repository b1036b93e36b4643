"""Throughput of the device Rayleigh integral (pairs/s) at a water-field-like size."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from babelbrain_amd import harness as H, RayleighAndBHTE as R
pts, ds = H._bowl_points(60e-3, 60e-3, 30, 0.0)
u0 = np.ones(len(ds), np.complex64)
for n in (1 << 18, 1 << 21, 1 << 23):
    rng = np.random.default_rng(0)
    rf = np.stack([rng.uniform(-50e-3, 50e-3, n), rng.uniform(-50e-3, 50e-3, n), rng.uniform(10e-3, 150e-3, n)], 1).astype(np.float32)
    R.ForwardSimple(2 * np.pi * 500e3 / 1500, pts, ds, u0, rf[:1000])
    t0 = time.time(); out = R.ForwardSimple(2 * np.pi * 500e3 / 1500, pts, ds, u0, rf); t1 = time.time()
    print('M=%d N=%d: kernel %.2f ms (%.1f Gpairs/s), wall %.2f s' % (len(ds), n, R.last_kernel_ms, len(ds) * n / R.last_kernel_ms / 1e6, t1 - t0))

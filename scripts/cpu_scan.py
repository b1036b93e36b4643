"""Oracle throughput vs OpenMP thread count on this host (for choosing the cpu_baseline setting)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from babelbrain_amd import harness as H
from oracle import oracle as O
a, k, info = H.make_problem('C3', N=(256, 256, 192), steps=12, stable_dt_fn=lambda ml, f, h, c: O.stable_dt(ml, f, True, h, c), accumulate_all_steps=True)
for nth in (16, 32, 64, 128, 256):
    out = O.StaggeredFDTD_3D_with_relaxation(*a, nthreads=nth, **k)
    s = out[-1]['stepLoopSeconds']
    print('threads %3d: %.1f Mvoxel-steps/s' % (nth, 256 * 256 * 192 * 12 / s / 1e6), flush=True)

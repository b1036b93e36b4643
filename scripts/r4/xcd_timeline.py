"""Experiment build -DBFD_EXP_XCD_CLOCK: throughput of a fluid launch over its duration (planes finished per 20 us bin) and the number of blocks in flight:
is there a slow start or a long tail? usage: BABELFDTD_HIP_LIB=$PWD/babelbrain_amd/libbabelfdtd_hip_xclk.so python scripts/r4/xcd_timeline.py [C3|C2]"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
cfg = sys.argv[1] if len(sys.argv) > 1 else 'C3'
sys.argv = ['bench.py', '--config', cfg, '--size', '512', '512', '512', '--steps', '10', '--warmup', '5', '--no-steady-warmup']
import bench
import numpy as np
from babelbrain_amd import _engine
args = bench.parse()
import torch
lib = _engine.load_library()
lib.bfd_debug_xcd_clock.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
w = bench.Workload(args, cfg, (512, 512, 512), 'weak', 0, 1, 0, None, lambda ml, f, h, a: _engine.stable_dt(ml, f, True, h, a), 10, 5, 0)
w.runner.run(30); torch.cuda.synchronize()
n = 1 << 17
w.runner.run(1); torch.cuda.synchronize()
for kind, name in ((0, 'stress_fluid'), (1, 'velocity_fluid')):
    st = np.zeros(n, np.uint64); en = np.zeros(n, np.uint64)
    lib.bfd_debug_xcd_clock(kind, n, st.ctypes.data, en.ctypes.data)
    nb = int((en != 0).sum())
    s = st[:nb].astype(np.float64); e = (en[:nb] >> np.uint64(4)).astype(np.float64)
    t0 = s.min(); s = (s - t0) / 100; e = (e - t0) / 100
    dur = e - s
    T = e.max(); bins = np.arange(0, T + 20, 20.0)
    done, _ = np.histogram(e, bins)
    started, _ = np.histogram(s, bins)
    inflight = [int(np.sum((s <= b) & (e > b))) for b in bins[:-1]]
    print('%s %s: %d blocks, launch %.1f us, block duration median %.1f us (p10 %.1f, p90 %.1f)' % (cfg, name, nb, T, np.median(dur), np.percentile(dur, 10), np.percentile(dur, 90)))
    print('   blocks finished per 20 us bin:', ' '.join(str(int(v)) for v in done))
    print('   blocks in flight at the bin start:', ' '.join(str(v) for v in inflight))
w.close()

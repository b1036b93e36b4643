"""Experiment build -DBFD_EXP_XCD_CLOCK (libbabelfdtd_hip_xclk.so): when does each XCD finish its part of the fluid launches, and does block b run
on XCD b & 7 (the assumption remap_block rests on)? Runs a 512^3 config one step at a time; per kernel class prints the XCC_ID histogram of the
slots b & 7 and the end times per XCD. usage: BABELFDTD_HIP_LIB=$PWD/babelbrain_amd/libbabelfdtd_hip_xclk.so python scripts/r4/xcd_clock.py [C3|C2|C1]"""
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
tc = w.eng.tile_counts()
w.runner.run(30)
torch.cuda.synchronize()
n = int(os.environ.get('XCLK_BLOCKS', 0)) or 1 << 17
acc = {0: [], 1: []}
for step in range(8):
    w.runner.run(1)
    torch.cuda.synchronize()
    for kind in (0, 1):
        st = np.zeros(n, np.uint64); en = np.zeros(n, np.uint64)
        lib.bfd_debug_xcd_clock(kind, n, st.ctypes.data, en.ctypes.data)
        used = en != 0
        nb = int(used.sum())
        st, en = st[:nb].astype(np.float64), en[:nb]
        xcc = (en & np.uint64(15)).astype(np.int64); end = (en >> np.uint64(4)).astype(np.float64)
        acc[kind].append((st, end, xcc))
for kind, name in ((0, 'stress_fluid'), (1, 'velocity_fluid')):
    st, end, xcc = acc[kind][-1]
    nb = len(st); t0 = st.min(); slot = np.arange(nb) & 7
    agree = float(np.mean(xcc == ((xcc[0] + slot) % 8)))
    hist = [[int(np.sum((slot == s) & (xcc == x))) for x in range(8)] for s in range(8)]
    print('%s %s: %d blocks, launch %.1f us; slot b & 7 -> XCC_ID of block 0 is %d; fraction of blocks on XCD (xcc0 + b) %% 8: %.4f' % (cfg, name, nb, (end.max() - t0) / 100, xcc[0], agree))
    print('   XCC_ID histogram of slot 0:', hist[0], ' slot 1:', hist[1])
    ends = np.array([[(a[1][a[2] == x].max() - a[0].min()) / 100 for x in range(8)] for a in acc[kind][2:]]).mean(axis=0)
    busy = np.array([[np.sum(a[1][a[2] == x] - a[0][a[2] == x]) / 100 for x in range(8)] for a in acc[kind][2:]]).mean(axis=0)
    print('   last block end per XCD (us): %s -> spread %.1f us = %.1f %% of the launch' % (' '.join('%.1f' % v for v in ends), ends.max() - ends.min(), 100 * (ends.max() - ends.min()) / ends.max()))
    print('   sum of block durations per XCD (ms): %s' % ' '.join('%.2f' % (v / 1e3) for v in busy))
w.close()

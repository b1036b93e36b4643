"""Random shapes, schedules, run lengths and monitors: the two-step kernels (bhte_step2g default, bhte_step2 with BFD_BHTE_KERNEL=1)
against one step per launch (BFD_BHTE_FUSE=0), every output bit for bit. usage: python scripts/r4/bhte_hunt.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from babelbrain_amd import RayleighAndBHTE as R

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ml = dict(Density=np.array([1000., 1896.5, 1041., 1100., 1850.]), SoS=np.array([1500., 2476., 1562., 1610., 2140.]), Attenuation=np.array([0., 81., 3.45, 20., 60.]),
          SpecificHeat=np.array([4178., 1313., 3630., 3391., 1793.]), Conductivity=np.array([0.6, 0.32, 0.51, 0.37, 0.31]), Perfusion=np.array([0., 10., 559., 106., 30.]),
          Absorption=np.array([0., 0.16, 0.85, 0.5, 0.2]), InitTemperature=np.array([37., 37., 37., 37., 37.]))
bad = 0
for c in range(cases):
    N = tuple(int(v) for v in (rng.integers(3, 200), rng.integers(3, 90), rng.integers(3, 70)))
    if c % 7 == 0: N = (int(rng.choice([63, 64, 65, 128, 129])), int(rng.choice([23, 24, 25, 48, 49])), int(rng.integers(3, 40)))     # tile edges
    nf = int(rng.integers(1, 4))
    onoff = [[int(rng.integers(0, 4)), int(rng.integers(0, 3))] for _ in range(nf)]
    if sum(a + b for a, b in onoff) == 0: onoff[0] = [1, 1]
    nS = int(rng.integers(1, 4)) * sum(a + b for a, b in onoff)
    mm = rng.integers(0, 5, N).astype(np.uint8)
    fields = (3.0e6 * rng.random((nf,) + N)).astype(np.float32)
    T0 = (37.0 + 8.0 * rng.random(N)).astype(np.float32)
    mpm = np.zeros(N, np.uint32); mpm[rng.integers(0, N[0]), rng.integers(0, N[1]), rng.integers(0, N[2])] = 1
    zrun = str(int(rng.choice([0, 1, 3, 5, 8, 13, 24, 40])))
    fm = int(rng.integers(1, 5))
    out = {}
    for fuse, kern in (('0', '0'), ('1', '0'), ('1', '1')):
        os.environ['BFD_BHTE_FUSE'] = fuse; os.environ['BFD_BHTE_KERNEL'] = kern
        if zrun != '0': os.environ['BFD_BHTE_ZRUN'] = zrun
        else: os.environ.pop('BFD_BHTE_ZRUN', None)
        out[fuse + kern] = R.BHTEMultiplePressureFields(fields, mm, ml, 4e-4, nS, onoff, int(rng.integers(0, N[1])) if fuse == '0' and kern == '0' else sl,
                                                        nFactorMonitoring=fm, dt=0.02, initT0=T0, MonitoringPointsMap=mpm) if (fuse, kern) != ('0', '0') else None
        if (fuse, kern) == ('0', '0'):
            sl = int(rng.integers(0, N[1]))
            out['00'] = R.BHTEMultiplePressureFields(fields, mm, ml, 4e-4, nS, onoff, sl, nFactorMonitoring=fm, dt=0.02, initT0=T0, MonitoringPointsMap=mpm)
    for k in ('10', '11'):
        ok = all(np.array_equal(a, b) for a, b in zip(out[k], out['00']))
        if not ok:
            bad += 1
            print('MISMATCH case %d kernel %s N=%s onoff=%s nS=%d zrun=%s' % (c, k, N, onoff, nS, zrun))
print('%d cases x 2 kernels, %d mismatches' % (cases, bad))
sys.exit(1 if bad else 0)

"""Round 6: random shapes, schedules, run lengths and monitors -- the S-step kernels (bhte_stepNg, S = 4 default and S = 3), the two-step kernels
(bhte_step2g with BFD_BHTE_STEPS=2, bhte_step2 with BFD_BHTE_KERNEL=1) against one step per launch (BFD_BHTE_FUSE=0), every output bit for bit.
usage: python scripts/r6/bhte_hunt.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from babelbrain_amd import RayleighAndBHTE as R

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ml = dict(Density=np.array([1000., 1896.5, 1041., 1100., 1850.]), SoS=np.array([1500., 2476., 1562., 1610., 2140.]), Attenuation=np.array([0., 81., 3.45, 20., 60.]),
          SpecificHeat=np.array([4178., 1313., 3630., 3391., 1793.]), Conductivity=np.array([0.6, 0.32, 0.51, 0.37, 0.31]), Perfusion=np.array([0., 10., 559., 106., 30.]),
          Absorption=np.array([0., 0.16, 0.85, 0.5, 0.2]), InitTemperature=np.array([37., 37., 37., 37., 37.]))
VARIANTS = {'one step': dict(BFD_BHTE_FUSE='0'), 'default (3 heating / 4 cooling)': {}, 'S=4': dict(BFD_BHTE_STEPS='4'), 'S=3': dict(BFD_BHTE_STEPS='3'), 'two steps': dict(BFD_BHTE_STEPS='2'),
            'two steps, round 3': dict(BFD_BHTE_KERNEL='1')}
bad = 0
for c in range(cases):
    N = tuple(int(v) for v in (rng.integers(3, 200), rng.integers(3, 90), rng.integers(3, 70)))
    if c % 7 == 0: N = (int(rng.choice([63, 64, 65, 128, 129])), int(rng.choice([19, 20, 21, 22, 23, 24, 25, 40, 44, 48, 49])), int(rng.integers(3, 40)))     # tile edges
    nf = int(rng.integers(1, 4))
    # stretches long enough for S-step passes in most cases, short ones in some
    onoff = [[int(rng.integers(0, 9)), int(rng.integers(0, 7))] for _ in range(nf)]
    if sum(a + b for a, b in onoff) == 0: onoff[0] = [1, 1]
    nS = int(rng.integers(1, 3)) * sum(a + b for a, b in onoff)
    mm = rng.integers(0, 5, N).astype(np.uint8)
    fields = (3.0e6 * rng.random((nf,) + N)).astype(np.float32)
    T0 = (37.0 + 8.0 * rng.random(N)).astype(np.float32)
    mpm = np.zeros(N, np.uint32)
    for q in range(int(rng.integers(1, 5))):          # 1-4 monitored points, some on faces / edges
        pt = [int(rng.integers(0, n)) for n in N]
        if q == 3: pt[int(rng.integers(0, 3))] = 0
        mpm[pt[0], pt[1], pt[2]] = q + 1
    zrun = str(int(rng.choice([0, 1, 3, 5, 8, 13, 24, 40])))
    fm = int(rng.choice([1, 2, 3, 4, 5, 7, 10]))
    sl = int(rng.integers(0, N[1])) if c % 5 else -1
    out = {}
    for name, env in VARIANTS.items():
        for k in ('BFD_BHTE_FUSE', 'BFD_BHTE_STEPS', 'BFD_BHTE_KERNEL'): os.environ.pop(k, None)
        os.environ.update(env)
        if zrun != '0': os.environ['BFD_BHTE_ZRUN'] = zrun
        else: os.environ.pop('BFD_BHTE_ZRUN', None)
        out[name] = R.BHTEMultiplePressureFields(fields, mm, ml, 4e-4, nS, onoff, sl, nFactorMonitoring=fm, dt=0.02, initT0=T0, MonitoringPointsMap=mpm)
    for name in VARIANTS:
        if name == 'one step': continue
        ok = all(np.array_equal(a, b) for a, b in zip(out[name], out['one step']))
        if not ok:
            bad += 1
            which = [i for i, (a, b) in enumerate(zip(out[name], out['one step'])) if not np.array_equal(a, b)]
            print('MISMATCH case %d kernel %s N=%s onoff=%s nS=%d zrun=%s fm=%d slice=%d outputs %s' % (c, name, N, onoff, nS, zrun, fm, sl, which), flush=True)
print('%d cases x %d kernels, %d mismatches' % (cases, len(VARIANTS) - 1, bad))
sys.exit(1 if bad else 0)

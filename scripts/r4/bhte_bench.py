"""BHTE kernel rate (two steps per launch) for A/B runs: BFD_BHTE_KERNEL=1 selects the round-3 kernel (bhte_step2), default = bhte_step2g.
usage: python scripts/r4/bhte_bench.py [N] [steps] [steps_on]   (N^3 voxels; kernel time = HIP events inside the C-ABI call)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from babelbrain_amd import RayleighAndBHTE as R

n = int(sys.argv[1]) if len(sys.argv) > 1 else 384
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
on = int(sys.argv[3]) if len(sys.argv) > 3 else steps // 2
N = (n, n, n)
h = 1102.515 / 500e3 / 6
rng = np.random.default_rng(0)
mm = np.zeros(N, np.uint8); mm[:, :, n // 4:n // 3] = 1; mm[:, :, n // 3:] = 2
ML = dict(Density=np.array([1000., 1896.5, 1041.]), SoS=np.array([1500., 2476., 1562.]), Attenuation=np.array([0., 81., 3.45]),
          SpecificHeat=np.array([4178., 1313., 3630.]), Conductivity=np.array([0.6, 0.32, 0.51]), Perfusion=np.array([0., 10., 559.]),
          Absorption=np.array([0., 0.16, 0.85]), InitTemperature=np.array([37., 37., 37.]))
P = (2e5 * rng.random(N, dtype=np.float32)).astype(np.float32)
R.BHTE(P[:64, :64, :64].copy(), mm[:64, :64, :64].copy(), ML, h, 4, 2, -1, dt=0.05)      # library load, first launches
best = None
for rep in range(3):
    out = R.BHTE(P, mm, ML, h, steps, on, -1, dt=0.05)
    ms = R.last_kernel_ms
    best = ms if best is None else min(best, ms)
vox = float(np.prod(N)) * steps
# algorithmic bytes per voxel-step of the two-step launch: heating (T r/w 8, dose r/w 8, q 4, id 1) / 2, cooling without q
byt = (21.0 * on + 17.0 * (steps - on)) / 2 / steps
print('BHTE %d^3 %d steps (%d heating) kernel=%s zrun=%s: %.2f ms -> %.0f Gvoxel-steps/s, %.2f of 8 TB/s on %.2f B per voxel-step; Tmax %.4f dose max %.4g'
      % (n, steps, on, os.environ.get('BFD_BHTE_KERNEL', 'g'), os.environ.get('BFD_BHTE_ZRUN', 'auto'), best, vox / best / 1e6, byt * vox / best / 1e6 / 8000, byt,
         float(out[0].max()), float(out[1].max())))

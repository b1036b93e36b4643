"""Throughput of the kernels either side of the FDTD path on one MI355X:
  * `ForwardSimple` (Rayleigh integral): source/point pairs per second -- a source plane (488 x 488 points) and a volume;
  * `BHTE`: voxel-steps per second and bytes per second on its algorithmic 21 B per voxel-step (T read + write, dose
    read + write, heat source, material id) against the 8 TB/s peak.
Kernel time = the HIP-event time the C-ABI call reports (uploads / downloads excluded)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from babelbrain_amd import RayleighAndBHTE as R

f, c = 500e3, 1500.0
tx = R.GenerateFocusTx(f, 50e-3, 50e-3, c)
cen, ds = tx['center'].astype(np.float32), tx['ds'].astype(np.float32)
u0 = np.ones(len(ds), np.complex64)
k = 2 * np.pi * f / c
h = 1102.515 / f / 6
for name, shape in (('source plane 488 x 488', (488, 488, 1)), ('volume 160 x 160 x 256', (160, 160, 256))):
    xs = (np.arange(shape[0]) - shape[0] / 2) * h; ys = (np.arange(shape[1]) - shape[1] / 2) * h; zs = 0.02 + np.arange(shape[2]) * h
    X, Y, Z = np.meshgrid(xs, ys, zs, indexing='ij')
    rf = np.stack([X.ravel(), Y.ravel(), Z.ravel()], 1).astype(np.float32)
    R.ForwardSimple(k, cen, ds, u0, rf[:1000])                          # warm-up (library load, first launch)
    t0 = time.time(); out = R.ForwardSimple(k, cen, ds, u0, rf); wall = time.time() - t0
    pairs = len(ds) * len(rf)
    print('Rayleigh %-24s %d sources x %d points = %.3e pairs: kernel %.2f ms -> %.0f Gpairs/s (call %.2f s); |p|max %.4g'
          % (name, len(ds), len(rf), pairs, R.last_kernel_ms, pairs / R.last_kernel_ms / 1e6, wall, float(np.abs(out).max())))

N = (384, 384, 384)
rng = np.random.default_rng(0)
mm = np.zeros(N, np.uint8); mm[:, :, 100:140] = 1; mm[:, :, 140:] = 2
ML = dict(Density=np.array([1000., 1896.5, 1041.]), SoS=np.array([1500., 2476., 1562.]), Attenuation=np.array([0., 81., 3.45]),
          SpecificHeat=np.array([4178., 1313., 3630.]), Conductivity=np.array([0.6, 0.32, 0.51]), Perfusion=np.array([0., 10., 559.]),
          Absorption=np.array([0., 0.16, 0.85]), InitTemperature=np.array([37., 37., 37.]))
P = (2e5 * rng.random(N, dtype=np.float32)).astype(np.float32)
steps, on = 200, 100
for fuse in ('1', '0'):
    os.environ['BFD_BHTE_FUSE'] = fuse
    t0 = time.time(); out = R.BHTE(P, mm, ML, h, steps, on, N[1] // 2, nFactorMonitoring=10, dt=0.05); wall = time.time() - t0
    vox = float(np.prod(N)) * steps
    ms = R.last_kernel_ms
    bpv = (sum(21.0 if heating else 17.0 for _, _, heating in R.bhte_pass_plan([0] * on + [-1] * (steps - on), 10, True)) / steps) if fuse == '1' else (21.0 * on + 17.0 * (steps - on)) / steps
    print('BHTE %dx%dx%d, %d steps (%d heating), %s: kernel %.1f ms -> %.0f Gvoxel-steps/s, %.2f B per voxel-step = %.2f of 8 TB/s (call %.1f s); Tmax %.3f'
          % (N + (steps, on, 'default path (four steps per pass)' if fuse == '1' else 'one step per launch', ms, vox / ms / 1e6, bpv,
             bpv * vox / ms / 1e6 / 8000, wall, float(out[0].max()))))

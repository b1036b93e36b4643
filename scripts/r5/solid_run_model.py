"""Model of lever (a)/(b) on the C2 medium at 512^3 (CPU, numpy): how many cells ride the solid-run kernels for a given
classification sub-tile, and how well the 128-byte lines of the solid-only arrays are used there.
Usage: python scripts/r5/solid_run_model.py [N]"""
import sys, numpy as np
sys.path.insert(0, '.')
from babelbrain_amd import harness as H

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
freq = 500e3
h = H.spatial_step(freq, 6)
m = H.skull_shell_map(N, N, N, h, H.PML_THICKNESS)            # [i, j, k]
solid = (m == 1)
print('cells', m.size / 1e6, 'M; solid', solid.sum() / 1e6, 'M')

def grown_any(a, tx, ty, tz, g=2):
    """a[i,j,k] bool -> per sub-tile flag: any True within the sub-tile grown by g cells"""
    n1, n2, n3 = a.shape
    # dilate by g along each axis (box), then block-reduce
    d = a.copy()
    for ax in range(3):
        acc = d.copy()
        for s in range(1, g + 1):
            sl_a = [slice(None)] * 3; sl_b = [slice(None)] * 3
            sl_a[ax] = slice(s, None); sl_b[ax] = slice(None, -s)
            acc[tuple(sl_b)] |= d[tuple(sl_a)]
            acc[tuple(sl_a)] |= d[tuple(sl_b)]
        d = acc
    b = d.reshape(n1 // tx, tx, n2 // ty, ty, n3 // tz, tz).any(axis=(1, 3, 5))
    return b

for (tx, ty, tz) in [(64, 8, 8), (64, 8, 4), (64, 4, 8), (64, 4, 4), (32, 8, 8), (32, 8, 4), (64, 8, 2), (64, 2, 8), (32, 4, 4), (64, 8, 1), (16, 8, 8)]:
    f = grown_any(solid, tx, ty, tz)
    cells = f.sum() * tx * ty * tz
    print(f'sub-tile {tx:3d}x{ty}x{tz}: {cells / 1e6:6.1f} M cells in solid runs ({solid.sum() / cells * 100:4.1f} % solid)')

# line usage of a solid-only array (value needed at solid cells) along x: 32 cells per 128 B line
s = solid.reshape(N // 32, 32, N, N)
lines = s.any(axis=1).sum()
print(f'solid-only array, full volume: {lines * 128 / 1e6:.1f} MB of lines touched for {solid.sum() * 4 / 1e6:.1f} MB needed '
      f'({solid.sum() * 4 / (lines * 128) * 100:.0f} % of each line used)')
# 64 B sectors (HBM burst granularity may be 64 B)
s = solid.reshape(N // 16, 16, N, N)
print(f'   at 64 B granularity: {s.any(axis=1).sum() * 64 / 1e6:.1f} MB')
s = solid.reshape(N // 8, 8, N, N)
print(f'   at 32 B granularity: {s.any(axis=1).sum() * 32 / 1e6:.1f} MB')

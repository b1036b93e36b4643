# round 4: Rayleigh kernel variants (LIBS="base <tag>"): tests, the error against the float64 oracle, the rate (scripts/next_rows_bench.py prints Rayleigh lines first)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r4_ray; mkdir -p $O
for tag in ${LIBS:-base}; do
  lib=babelbrain_amd/libbabelfdtd_hip.so; [ $tag != base ] && lib=babelbrain_amd/libbabelfdtd_hip_$tag.so
  export BABELFDTD_HIP_LIB=$PWD/$lib
  echo "== $tag"
  timeout 900 python -m pytest tests/test_rayleigh_gpu.py ${STUDY:+tests/test_rayleigh_study_gpu.py} -x -q 2>&1 | tail -1
  python - <<'PY'
import numpy as np, sys
sys.path.insert(0, '.')
from babelbrain_amd import RayleighAndBHTE as R, harness as H
from oracle import rayleigh_oracle as RO
rng = np.random.default_rng(5)
pts, ds = H._bowl_points(60e-3, 55e-3, 30, 0.0)
u0 = (rng.normal(size=len(ds)) + 1j * rng.normal(size=len(ds))).astype(np.complex64)
rf = np.stack([rng.uniform(-40e-3, 40e-3, 20000), rng.uniform(-40e-3, 40e-3, 20000), rng.uniform(20e-3, 160e-3, 20000)], 1).astype(np.float32)
for kim in (0.0, -4.5):
    k = np.array(2 * np.pi * 700e3 / 1500.0 + 1j * kim).astype(np.complex64)
    got = R.ForwardSimple(k, pts.astype(np.float32), ds.astype(np.float32), u0, rf); ref = RO.ForwardSimple(k, pts.astype(np.float32), ds.astype(np.float32), u0, rf)
    e = np.linalg.norm(got.astype(np.complex128) - ref) / np.linalg.norm(ref)
    print('   rel-L2 against the float64 oracle (Im k = %g): %.3e, max |diff| / max |ref| %.3e' % (kim, e, np.abs(got - ref).max() / np.abs(ref).max()))
PY
  for i in 1 2; do timeout 300 python scripts/next_rows_bench.py 2>&1 | grep Rayleigh; done
done | tee $O/rates_${OUT:-run}.txt

# round 4: the CPU oracle (oracle/fdtd_oracle.c) under AddressSanitizer + UndefinedBehaviorSanitizer (CPU only: the GPU pool refuses sanitizer runs).
# A scratch copy of the tree gets the instrumented library; the physics KATs and the host-logic tests run against it with the output uncaptured.
# Round 4 result (12.7 min, 4 threads): 18 passed, no "runtime error" / "AddressSanitizer" line.
set -e
R=$(cd "$(dirname "$0")/../.." && pwd); T=${TMPDIR:-/tmp}/oracle_asan; rm -rf $T; mkdir -p $T/repo
(cd $R && tar --exclude=gpurun_out --exclude=.git --exclude='*.o' --exclude=profiles -cf - .) | (cd $T/repo && tar xf -)
gcc -O1 -g -fPIC -std=gnu11 -ffp-contract=off -fno-fast-math -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer -shared -o $T/repo/oracle/libfdtd_oracle.so $R/oracle/fdtd_oracle.c -lm
touch $T/repo/oracle/libfdtd_oracle.so
cd $T/repo
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 BABEL_ORACLE_THREADS=4 \
  python -m pytest tests/test_oracle_physics.py tests/test_host_logic.py -x -q -s > $T/run.log 2>&1 || true
tail -n 2 $T/run.log; echo "sanitizer reports: $(grep -c 'runtime error\|AddressSanitizer' $T/run.log)"

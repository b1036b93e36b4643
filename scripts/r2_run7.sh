cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_run7; mkdir -p $O
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_dft_gpu.py -m gpu -q -x -k "streamed or in_loop or dft" -o faulthandler_timeout=300 > $O/tests.log 2>&1; tail -5 $O/tests.log
for v in "--cases 9 36 --pml 24" "--cases 9 36 --depth-mm 35" ; do echo "== $v"; timeout 600 python scripts/rayleigh_study_sweep.py $v 2>/dev/null | cut -c1-150; done
timeout 2400 python scripts/rayleigh_study_sweep.py --zadj 0 -10 10 --out $O/study_all.json > $O/study_all.log 2>&1; tail -3 $O/study_all.log

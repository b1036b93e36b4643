cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_run4; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x -k "not c5_1024 and not c4_h317 and not c2_full" --durations=5 -o faulthandler_timeout=600 > $O/tests.log 2>&1; tail -8 $O/tests.log

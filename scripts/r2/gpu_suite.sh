cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_run9; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x --durations=12 -o faulthandler_timeout=1500 > $O/tests.log 2>&1; tail -22 $O/tests.log

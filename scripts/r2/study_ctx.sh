cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_ctx; mkdir -p $O
timeout 1500 python scripts/rayleigh_study_sweep.py --tx CTX_500 --zadj 0 -10 10 --out $O/study_ctx.json > $O/study_ctx.log 2>&1; cat $O/study_ctx.log | cut -c1-170
timeout 600 python scripts/rayleigh_study_sweep.py --cases 18 19 45 72 99 126 --zadj 10 2>/dev/null | cut -c1-170

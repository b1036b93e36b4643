cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2_run5; mkdir -p $O
timeout 1200 python scripts/rayleigh_study_sweep.py --zadj 0 --every 3 --out $O/study_taylor.json > $O/study_taylor.log 2>&1; tail -20 $O/study_taylor.log
BABELFDTD_HIP_LIB=$PWD/babelbrain_amd/libbabelfdtd_hip_holberg.so timeout 1200 python scripts/rayleigh_study_sweep.py --zadj 0 --every 3 --out $O/study_holberg.json > $O/study_holberg.log 2>&1; tail -20 $O/study_holberg.log

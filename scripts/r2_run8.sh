cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for a in 1 0 0.25 ; do echo "== alpha scale $a"; BFD_CPML_ALPHA_SCALE=$a timeout 900 python scripts/rayleigh_study_sweep.py --zadj 0 --cases 9 10 12 36 40 41 63 66 90 117 2>/dev/null | cut -c1-140; done

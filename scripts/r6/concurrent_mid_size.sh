# round 6: velocity_solid beside velocity_fluid on a side stream (BFD_CONCURRENT=1) on mid-size grids, where one launch of the solid runs is a few rounds of workgroups
cd $GRAFT_REPO_ROOT
for cfg in "C2" "C2 --size 320 320 320" "C2 --size 384 384 384"; do
  VAR=BFD_CONCURRENT REPS=2 bash scripts/r6/ab_env.sh --config $cfg --steps 200 --warmup 40 --no-kernel-pass 2>&1 | sed "s/^/$cfg: /"
done

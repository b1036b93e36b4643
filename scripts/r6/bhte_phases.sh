# round 6: heating-only and cooling-only rates of the bio-heat kernels by steps per pass. usage: bash scripts/r6/bhte_phases.sh [sizes]
cd $GRAFT_REPO_ROOT
for n in ${@:-320 512}; do
  for on in 0 200; do
    for st in 2 3 4; do
      BFD_BHTE_STEPS=$st python scripts/r4/bhte_bench.py $n 200 $on 2>&1 | grep "^BHTE" | sed "s/^BHTE/steps=$st on=$on BHTE/" | cut -c1-140
    done
  done
done

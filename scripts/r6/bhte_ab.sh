# round 6: steps per pass of the bio-heat solver (BFD_BHTE_STEPS = 2: bhte_step2g; 3 / 4: bhte_stepNg), same box, in turn. usage: bash scripts/r6/bhte_ab.sh [sizes]
cd $GRAFT_REPO_ROOT
for n in ${@:-320 384 512}; do
  for rep in 1 2; do
    for st in 2 3 4; do
      BFD_BHTE_STEPS=$st python scripts/r4/bhte_bench.py $n 200 100 2>&1 | grep "^BHTE" | sed "s/^BHTE/steps=$st BHTE/"
    done
  done
done
for z in 16 24 32 48 64; do BFD_BHTE_STEPS=4 BFD_BHTE_ZRUN=$z python scripts/r4/bhte_bench.py 320 200 100 2>&1 | grep "^BHTE" | sed "s/^BHTE/steps=4 BHTE/"; done
for z in 16 24 32 48 64; do BFD_BHTE_STEPS=3 BFD_BHTE_ZRUN=$z python scripts/r4/bhte_bench.py 320 200 100 2>&1 | grep "^BHTE" | sed "s/^BHTE/steps=3 BHTE/"; done

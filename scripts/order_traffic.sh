# HBM traffic per kernel (deterministic, unlike timings) for run-order candidates
for v in 3 4; do for o in 2 6 7 9; do
echo "=== variant $v order $o"
BFD_RUN_ORDER=$o bash scripts/pmc_traffic.sh ord_${v}_$o --config C3 --size 512 512 512 --variant $v 2>/dev/null | grep -E "^== void|HBM bytes" | grep -v record
done; done

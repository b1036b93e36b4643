# kernel + memory-copy trace of the one-process split (two slabs of C3 on one GPU: the halo copies are device copies on the side
# stream): how much of the copies' time lies under kernels of the other stream
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r3_group_trace; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -o t -- python3 scripts/r3/group_bench.py --config C3 --slabs 2 --steps 40 --warmup 10 > $O/group_bench.txt 2>$O/err.txt
cat $O/group_bench.txt
python3 scripts/r3/group_overlap_summary.py $O

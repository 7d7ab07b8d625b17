cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c3prof -o c3 -- python3 $GRAFT_REPO_ROOT/tools/c3_time.py --reps 4 > $GRAFT_REPO_ROOT/gpurun_out/c3_p.log 2>&1
cd $GRAFT_REPO_ROOT; grep "^{" gpurun_out/c3_p.log; find gpurun_out/c3prof -name "*kernel_stats.csv" | xargs cut -c1-150 | head -8
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c5prof3 -o c5 -- python3 $GRAFT_REPO_ROOT/tools/c5_time.py --reps 2 > $GRAFT_REPO_ROOT/gpurun_out/c5_d.log 2>&1
cd $GRAFT_REPO_ROOT; grep "^{" gpurun_out/c5_d.log; find gpurun_out/c5prof3 -name "*kernel_stats.csv" | xargs cut -c1-150 | head -8
for k in and2 mix3 phrase; do python3 tools/c5_time.py --reps 2 --kind $k 2>&1 | grep "^{"; done

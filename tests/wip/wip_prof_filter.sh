cd /tmp && export TMPDIR=/tmp
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_filter -o filter -- python3 $GRAFT_REPO_ROOT/tests/wip/wip_perf_filter.py > $GRAFT_REPO_ROOT/gpurun_out/prof_filter.log 2>&1
tail -12 $GRAFT_REPO_ROOT/gpurun_out/prof_filter.log

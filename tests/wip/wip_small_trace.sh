cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --hip-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_small -o small -- python3 $GRAFT_REPO_ROOT/tests/wip/wip_small_trace.py > $GRAFT_REPO_ROOT/gpurun_out/prof_small.log 2>&1

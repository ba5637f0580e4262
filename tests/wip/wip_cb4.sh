cd /tmp && export TMPDIR=/tmp
for w in cb_userid; do
timeout -k 10 280 rocprofv3 --kernel-trace --hip-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$w -o $w -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --no-cpu-baseline --steps 5 --warmup 2 > $GRAFT_REPO_ROOT/gpurun_out/prof_$w.log 2>&1 || exit 1
done

#!/bin/bash
# Kernel traces of the ClickBench group-bys on the GPU box.  Usage: bash scripts/profile_cb.sh <set name> [workloads...]
# Summaries land in gpurun_out/prof_<set>/; copy the ones to keep into profiles/.
set -e
SET=$1; shift
WORKLOADS=${@:-cb_userid cb_phrase cb_adv}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_$SET
mkdir -p $O
for w in $WORKLOADS; do
  ROWS=""
  if [ $w = cb_adv ]; then ROWS="--rows 1000000000"; fi
  timeout -k 10 300 python3 $R/bench.py --steps 5 --warmup 3 --workload $w $ROWS --no-cpu-baseline --no-also > $O/${w}_plain_bench_line.json 2> $O/${w}_plain.err
  echo "plain $w: $(cat $O/${w}_plain_bench_line.json | head -c 600)"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$w -o $w -- python3 $R/bench.py --steps 5 --warmup 3 --workload $w $ROWS --no-cpu-baseline --no-also > $O/${w}_bench_line.json 2> $O/${w}.err
  echo "done trace $w"
done

#!/bin/bash
# HBM traffic of every kernel of the ClickBench group-bys: one kernel trace, then FETCH_SIZE and WRITE_SIZE counter passes
# in runs of their own (counters and tracing are never combined).
#   bash scripts/profile_cb_pmc.sh <set name, e.g. r3_a> [workloads...]
# Results land in gpurun_out/prof_<set>/; scripts/pmc_kernels_to_json.py turns them into profiles/<set>_pmc_traffic.json.
set -e
SET=$1; shift
WORKLOADS=${@:-cb_userid cb_phrase}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_$SET
mkdir -p $O
for w in $WORKLOADS; do
  ROWS=""
  if [ $w = cb_adv ] || [ $w = cb_q1 ]; then ROWS="--rows 1000000000"; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$w -o $w -- python3 $R/bench.py --steps 5 --warmup 3 --workload $w $ROWS --no-cpu-baseline --no-also > $O/${w}_bench_line.json 2> $O/${w}.err
  echo "done trace $w: $(head -c 400 $O/${w}_bench_line.json)"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${c}_$w -o pmc_$w -- python3 $R/bench.py --steps 3 --warmup 3 --workload $w $ROWS --no-cpu-baseline --no-also > $O/pmc_${c}_${w}_bench_line.json 2> $O/pmc_${c}_${w}.err
    echo "done pmc $c $w"
  done
done
python3 $R/scripts/pmc_kernels_to_json.py $O $O/${SET}_pmc_traffic.json $WORKLOADS

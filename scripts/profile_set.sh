#!/bin/bash
# One profile set on the GPU box: rocprofv3 kernel traces of the TPC-H workloads at SF100 and FETCH_SIZE passes at
# 128 M rows (counters in runs of their own).  Usage (from the repository root on the box):
#   bash scripts/profile_set.sh <set name, e.g. r2_a> [workloads...]
# Summaries land in gpurun_out/prof_<set>/; copy the ones to keep into profiles/ (scripts/pmc_to_json.py makes the
# traffic file from the counter passes).
set -e
SET=$1; shift
WORKLOADS=${@:-q1 q1_codes q6}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_$SET
mkdir -p $O
for w in $WORKLOADS; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$w -o $w -- python3 $R/bench.py --steps 5 --warmup 2 --workload $w --no-cpu-baseline --no-also > $O/${w}_bench_line.json 2> $O/${w}.err
  echo "done trace $w"
done
for w in $WORKLOADS; do
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_$w -o pmc_$w -- python3 $R/bench.py --steps 3 --warmup 1 --rows 128000000 --workload $w --no-cpu-baseline --no-also > $O/pmc_${w}_bench_line.json 2> $O/pmc_${w}.err
  echo "done pmc $w"
done
find $O -name "*stats.csv" -o -name "*counter_collection.csv" | head -40

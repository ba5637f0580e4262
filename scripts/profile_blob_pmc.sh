#!/bin/bash
# HBM traffic of the blob import's kernels: FETCH_SIZE and WRITE_SIZE counter passes, a run each, no tracing.
#   bash scripts/profile_blob_pmc.sh <set name, e.g. r3_e> [portions]
# scripts/pmc_kernels_to_json.py turns gpurun_out/prof_<set>/ into <set>_pmc_traffic.json (workload "blob<portions>").
set -e
SET=$1
PORTIONS=${2:-32}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_$SET
mkdir -p $O
W=blob$PORTIONS
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${c}_$W -o pmc_$W -- python3 $R/scripts/bench_blob.py $PORTIONS > $O/pmc_${c}_${W}.txt 2> $O/pmc_${c}_${W}.err
  echo "done pmc $c $W"
done
python3 $R/scripts/pmc_kernels_to_json.py $O $O/${SET}_pmc_traffic.json $W > /dev/null

#!/bin/bash
# Kernel trace of a blob import on the GPU box.  Usage: bash scripts/profile_blob.sh <set name> [portions]
# The summary lands in gpurun_out/prof_<set>/; copy what is to be kept into profiles/.
set -e
SET=$1
PORTIONS=${2:-32}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof_$SET
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/blob -o blob -- python3 $R/scripts/bench_blob.py $PORTIONS > $O/blob_bench.txt 2> $O/blob.err
echo "done trace blob"

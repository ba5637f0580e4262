#!/bin/bash
# Decode kernels of a blob import per kind of column (rocprofv3 kernel statistics of scripts/bench_blob.py with one column kept).
# Run on the GPU box from the repository root: bash scripts/blob_columns.sh [portions]
set -o pipefail
N=${1:-32}
export PYTHONPATH=.
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out/blobcols
for col in l_shipdate l_quantity l_extendedprice l_tax l_returnflag; do
    echo "== $col"
    BLOB_COLUMNS=$col timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/blobcols/$col -o p -- python3 scripts/bench_blob.py "$N" > gpurun_out/blobcols/$col.log 2>&1 || { tail -5 gpurun_out/blobcols/$col.log; exit 1; }
    grep -E "decode kernel|portions" gpurun_out/blobcols/$col.log
    f=$(find gpurun_out/blobcols/$col -name '*kernel_stats.csv' | head -1)
    grep -E "Lz4" "$f" | cut -d, -f1-4,6,7
done

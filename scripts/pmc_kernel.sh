#!/bin/bash
# Hardware counters of the kernels of one workload (a run of its own: no tracing).
#   bash scripts/pmc_kernel.sh <tag> "<counters>" <program and arguments...>
# The rows of the kernels of this library are summed per kernel and counter and printed; the raw files stay in gpurun_out/pmc_<tag>/.
TAG=$1; COUNTERS=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmc_$TAG
mkdir -p $O
timeout -s KILL 120 rocprofv3 --pmc $COUNTERS --output-format csv -d $O -o p -- "$@" > $O/out.txt 2> $O/err.txt
python3 - <<PY
import csv, glob, collections
files = glob.glob("$O/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for fn in files:
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].split("<")[0].split("::")[-1]
        if "at::" in r["Kernel_Name"] or k.startswith("__amd"): continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
for k in acc:
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f"   {c:28s} total {v:16.0f}  rows {n[(k, c)]}")
PY

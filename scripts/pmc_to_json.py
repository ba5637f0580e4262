#!/usr/bin/env python3
"""Turns the FETCH_SIZE passes of scripts/profile_set.sh into the traffic file bench.py reads.

    python scripts/pmc_to_json.py gpurun_out/prof_<set> profiles/<set>_pmc_traffic.json [workloads...]

FETCH_SIZE is in KiB and, on gfx950, reports half of the bytes of 16-byte-per-lane streaming reads
(MI355X_MICROARCH.md, HBM section): corrected bytes = 2 * 1024 * FETCH_SIZE.  Averages over the launches of the scan
kernel built at run time for the program (ssa_spec_*), whose algorithmic bytes come from the bench line of the pass."""
import csv
import glob
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
workloads = sys.argv[3:] or ["q1", "q1_codes", "q6"]
out = {"_about": "rocprofv3 --pmc FETCH_SIZE (own pass, no tracing) around `bench.py --steps 3 --warmup 1 --rows 128000000 --workload W "
                 "--no-cpu-baseline --no-also`; FETCH_SIZE is in KiB and, on gfx950, reports half of the bytes of 16-byte-per-lane "
                 "streaming reads, so corrected_bytes = 2 * 1024 * FETCH_SIZE. Averages over the launches of the scan kernel built at "
                 "run time for the program."}
for w in workloads:
    files = glob.glob(os.path.join(src, f"pmc_{w}", "**", "*counter_collection.csv"), recursive=True)
    if not files:
        out[w] = {"note": "no counter pass found"}
        continue
    vals = []
    kernel = None
    for row in csv.DictReader(open(files[0])):
        if row.get("Counter_Name") == "FETCH_SIZE" and row.get("Kernel_Name", "").startswith("ssa_spec_"):
            vals.append(float(row["Counter_Value"]))
            kernel = row["Kernel_Name"]
    line = json.loads(open(os.path.join(src, f"pmc_{w}_bench_line.json")).read().strip().splitlines()[-1])
    algo = line["roofline"]["algorithmic_bytes_per_launch"]
    if not vals:
        out[w] = {"note": "no launch of a kernel built for the program in the pass"}
        continue
    # the dispatches of one launch are summed by the tool per dispatch id: one row per launch
    kib = sum(vals) / len(vals)
    out[w] = {"kernel": kernel, "rows": line["config"]["rows_per_gpu"], "launches": len(vals), "fetch_size_kib": kib,
              "corrected_bytes": 2 * 1024 * kib, "algorithmic_bytes": algo, "ratio_to_algorithmic": 2 * 1024 * kib / algo}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))

#!/usr/bin/env python3
"""Per-kernel HBM traffic from the counter passes of scripts/profile_cb_pmc.sh.

    python scripts/pmc_kernels_to_json.py gpurun_out/prof_<set> profiles/<set>_pmc_traffic.json [workloads...]

For every kernel of a workload: launches seen, average FETCH_SIZE and WRITE_SIZE per launch (KiB, as the tool reports them)
and the bytes they stand for.  On gfx950 FETCH_SIZE tallies a 128-byte request of a wide streaming read as 64 bytes, so
`read_bytes_streaming` doubles it (right for 16-byte-per-lane loads, an upper bound for narrow or scattered ones, for which
`read_bytes_raw` is the lower bound); WRITE_SIZE is taken as it is.  The last warm-up-free launches are what counts: the
first launches of a step (table sizing, kernel builds) are in the averages too, so each kernel also lists its median."""
import csv
import glob
import json
import os
import statistics
import sys

src, dst = sys.argv[1], sys.argv[2]
workloads = sys.argv[3:] or ["cb_userid", "cb_phrase"]


def short(name):
    name = name.replace("(anonymous namespace)::", "").split("(")[0].split("<")[0]
    return name.split("::")[-1]


out = {"_about": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, one run each, no tracing, around `bench.py --steps 3 --warmup 3 "
                 "--workload W --no-cpu-baseline --no-also`; KiB per launch as reported; read_bytes_streaming = 2 * 1024 * FETCH_SIZE "
                 "(gfx950 counts a 128-byte streaming request as 64), read_bytes_raw = 1024 * FETCH_SIZE, write_bytes = 1024 * WRITE_SIZE"}
for w in workloads:
    per = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        files = glob.glob(os.path.join(src, f"pmc_{counter}_{w}", "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        vals = {}
        for row in csv.DictReader(open(files[0])):
            if row.get("Counter_Name") != counter:
                continue
            k = short(row.get("Kernel_Name", ""))
            if k.startswith("void at::") or "at::native" in row.get("Kernel_Name", ""):
                continue
            # (one row per dispatch and counter dimension: sum the dimensions of a dispatch)
            vals.setdefault(k, {}).setdefault(row.get("Dispatch_Id"), 0.0)
            vals[k][row.get("Dispatch_Id")] += float(row["Counter_Value"])
        for k, d in vals.items():
            v = list(d.values())
            e = per.setdefault(k, {})
            e[counter.lower() + "_kib_avg"] = sum(v) / len(v)
            e[counter.lower() + "_kib_median"] = statistics.median(v)
            e["launches"] = len(v)
    for k, e in per.items():
        if "fetch_size_kib_median" in e:
            e["read_bytes_raw"] = 1024 * e["fetch_size_kib_median"]
            e["read_bytes_streaming"] = 2048 * e["fetch_size_kib_median"]
        if "write_size_kib_median" in e:
            e["write_bytes"] = 1024 * e["write_size_kib_median"]
    line = None
    p = os.path.join(src, f"{w}_bench_line.json")
    if os.path.exists(p):
        try:
            line = json.loads(open(p).read().strip().splitlines()[-1])
        except Exception:
            line = None
    entry = {"kernels": per}
    if line:
        entry["rows"] = line.get("config", {}).get("rows_per_gpu")
        entry["ms_per_step"] = line.get("ms_per_step")
        entry["algorithmic_bytes_per_step"] = line.get("roofline", {}).get("algorithmic_bytes_per_launch")
    out[w] = entry
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])

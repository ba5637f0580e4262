#!/usr/bin/env python3
"""Scan-pushdown benchmark: rows/s of the SSA program executor over an HBM-resident lineitem.

    python bench.py --gpus N --steps K --warmup W [--workload q1|q1_codes|q6|cb_*] [--rows R]

One step = one pass of the program (TPC-H Q1 by default: filter, two computed columns, nine aggregates
grouped by (l_returnflag, l_linestatus)) over this rank's SF100 lineitem shard, result on the host.  With
N > 1 (one process per GPU, launched by torch.distributed.run) every rank scans its own SF100-sized
row range and the partial aggregates are merged through one RCCL all-gather plus a merge program, all
inside the timed region; `value` is the rows all ranks scanned divided by the slowest rank's time.

The cb_* workloads are the ClickBench scan-aggregate programs over a synthetic 100 M-row `hits` (ydb_amd/clickbench.py);
their high-cardinality group-bys leave the result table in HBM on one GPU (the row count is read back).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E peak
PMC_TRAFFIC_FILE = "r1_d_pmc_traffic.json"  # FETCH_SIZE passes of the scan kernels, as ratios to the algorithmic bytes


def cpu_baseline(workload: str, seconds_target: float = 15.0):
    """Times the CPU oracle (a port of the reference's op-at-a-time execution) on a bounded sample."""
    try:
        from oracle import ssa_oracle
    except Exception as e:  # oracle library not built
        return {"value": None, "unit": "rows/s", "cores": 0, "kind": "port", "sample": f"unavailable: {e}"}
    from ydb_amd import tpch
    import numpy as np

    batch_rows = 1_000_000  # the reference hands the program batches of about this size
    if workload.startswith("cb_"):
        from ydb_amd import clickbench
        prog, cols = clickbench.programs()[workload], clickbench.COLUMNS[workload][0]
        batch = clickbench.DeviceHits(batch_rows, columns=cols).host(cols)
        resolver = clickbench.HITS_RESOLVER
    else:
        codes = workload == "q1_codes"
        if workload == "q6":
            prog, cols = tpch.q6_program(), tpch.Q6_COLUMNS
        else:
            prog, cols = tpch.q1_program(codes), (tpch.Q1_COLUMNS_CODES if codes else tpch.Q1_COLUMNS_UTF8)
        li = tpch.lineitem_host(0, batch_rows)
        batch = {c: li[c] for c in cols}
        resolver = tpch.LINEITEM_RESOLVER
    orc = ssa_oracle.Oracle(resolver, prog)
    hb = orc.import_batch(batch)
    orc.apply_imported(hb)  # warm
    t0 = time.perf_counter()
    n = 0
    while True:
        orc.apply_imported(hb)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds_target or n >= 400:
            break
    out = {"value": n * batch_rows / dt, "unit": "rows/s", "cores": 1, "kind": "port",
           "sample": f"{n} x {batch_rows}-row batches of the same synthetic table, {dt:.1f} s, one thread"}
    # The reference runs one batch per Conveyor worker thread: the same loop on every core of this box's CPU share
    # (the oracle is called through ctypes, which releases the interpreter lock), each thread with its own program.
    import threading
    cores = max(1, min(16, os.cpu_count() or 1))
    per_thread = max(1, int(n * min(1.0, 8.0 / max(dt, 1e-3))))
    workers = []
    for _ in range(cores):
        o = ssa_oracle.Oracle(resolver, prog)
        workers.append((o, o.import_batch(batch)))
    start = threading.Barrier(cores + 1)

    def loop(o, h):
        start.wait()
        for _ in range(per_thread):
            o.apply_imported(h)

    threads = [threading.Thread(target=loop, args=w) for w in workers]
    for t in threads:
        t.start()
    start.wait()
    t0 = time.perf_counter()
    for t in threads:
        t.join()
    pdt = time.perf_counter() - t0
    out["all_cores"] = {"value": cores * per_thread * batch_rows / pdt, "unit": "rows/s", "cores": cores,
                        "sample": f"{cores} threads x {per_thread} batches, {pdt:.1f} s"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="q1", choices=["q1", "q1_codes", "q6", "cb_q1", "cb_q2", "cb_q3", "cb_q6", "cb_adv",
                                                         "cb_userid", "cb_phrase", "cb_q20"])
    ap.add_argument("--rows", type=int, default=0, help="rows per GPU (default: TPC-H SF100 lineitem; ClickBench hits for cb_*)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="every rank uses cuda:0 and the exchange runs over gloo (checks the N > 1 path on a 1-GPU box)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from ydb_amd import tpch
    from ydb_amd.shard import ShardedScan

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the scan kernels have no CPU fallback)")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    gather_dev = dev
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            gather_dev = torch.device("cpu")
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    rows = args.rows or tpch.SF100_ROWS
    codes = args.workload == "q1_codes"
    clickbench_run = args.workload.startswith("cb_")
    resolver = tpch.LINEITEM_RESOLVER
    table = "SF100-sized lineitem shard"
    if clickbench_run:
        from ydb_amd import clickbench
        rows = args.rows or clickbench.HITS_ROWS
        prog = clickbench.programs()[args.workload]
        cols, bpr = clickbench.COLUMNS[args.workload]
        name = "ClickBench " + args.workload[3:]
        resolver = clickbench.HITS_RESOLVER
        table = "hits-sized shard"
        li = clickbench.DeviceHits(rows, device=local_rank, row0=rank * rows, columns=cols)
        if args.workload == "cb_phrase":
            bpr = 4 + li.phrase_bytes / rows
        if args.workload == "cb_q20":
            bpr = 4 + li.url_bytes / rows
    elif args.workload == "q6":
        prog, cols, bpr = tpch.q6_program(), tpch.Q6_COLUMNS, tpch.Q6_BYTES_PER_ROW
        name = "TPC-H Q6"
    else:
        prog = tpch.q1_program(codes)
        cols = tpch.Q1_COLUMNS_CODES if codes else tpch.Q1_COLUMNS_UTF8
        bpr = tpch.Q1_BYTES_PER_ROW_CODES if codes else tpch.Q1_BYTES_PER_ROW_UTF8
        name = "TPC-H Q1" + (" (1-byte flag codes)" if codes else " (utf8 flags)")
    if not clickbench_run:
        li = tpch.DeviceLineitem(rows, device=local_rank, row0=rank * rows,
                                 utf8_flags=(args.workload == "q1"), code_flags=codes)
    scan = ShardedScan(prog, resolver, device=local_rank)
    batch = li.batch(cols)
    # a result of millions of groups stays in HBM on one GPU (its row count is read back)
    on_device = args.workload in ("cb_userid", "cb_phrase") and world == 1

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    result = None
    # The executor builds the step program's kernel at the first large scan and settles how it runs it (register-held
    # groups on or off) on the second: with fewer than two warm-up steps asked for, those passes run here, untimed,
    # like the generation of the table.
    for _ in range(max(0, 2 - args.warmup)):
        scan.run(batch, gather_device=gather_dev, keep_on_device=on_device)
    for _ in range(args.warmup):
        result = scan.run(batch, gather_device=gather_dev, keep_on_device=on_device)
    scan.local.EnableProfile(True)
    scan.local.ReadProfile()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        result = scan.run(batch, gather_device=gather_dev, keep_on_device=on_device)
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms, launches, algo_bytes = scan.local.ReadProfile()
    built, build_ms, spec_launches = scan.local.SpecializationStats()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=gather_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        # HBM traffic of the scan kernel: measured in a separate rocprofv3 --pmc FETCH_SIZE pass (profiles/), kept as
        # a ratio to the algorithmic bytes of the same launch; null when no such measurement is committed.
        traffic = None
        try:
            with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", PMC_TRAFFIC_FILE)) as f:
                pmc = json.load(f).get(args.workload)
            if pmc and "note" not in pmc and launches:
                traffic = pmc["ratio_to_algorithmic"] * (algo_bytes / launches)
        except OSError:
            pass
        from ydb_amd.shard import last_group_by
        keyed = bool(last_group_by(prog).keys)
        groups = result.rows if on_device else (len(next(iter(result.values())).valid) if result else 0)
        total_rows = rows * world * args.steps
        avg_kernel_s = (kernel_ms / 1e3) / max(1, launches)
        achieved = (algo_bytes / max(1, launches)) / avg_kernel_s / 1e9 if launches else 0.0
        out = {
            "metric": "rows/sec scanned + % HBM roofline, TPC-H Q1/Q6 SF100 at 1/2/4/8 MI355X",
            "value": total_rows / elapsed,
            "unit": "rows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{name}, {table} per GPU ({rows} rows), whole shard as one HBM-resident batch",
                       "rows_per_gpu": rows, "algorithmic_bytes_per_row": bpr,
                       "groups": groups,
                       "kernels_built_at_run_time": built, "kernel_build_ms_before_the_timed_steps": build_ms,
                       "parallelism": f"row-range shards x{world}, partial aggregates merged by " +
                                      ("hash-partitioned point-to-point exchange" if world > 1 and groups > ShardedScan.ALL_TO_ALL_GROUPS
                                       else "RCCL all-gather") + " + merge program"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         # the scan kernel built at run time for this program (same source as the generic
                         # ScanKeyedKernel / ScanNoKeyKernel, the program's tables as constants) when one ran
                         "kernel": (("ssa_spec_keyed" if keyed else "ssa_spec_nokey") if spec_launches
                                    else ("ScanKeyedKernel" if keyed else "ScanNoKeyKernel")),
                         "avg_kernel_ms": 1e3 * avg_kernel_s, "launches": launches,
                         "algorithmic_bytes_per_launch": algo_bytes // max(1, launches)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

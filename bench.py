#!/usr/bin/env python3
"""Scan-pushdown benchmark: rows/s of the SSA program executor over an HBM-resident lineitem.

    python bench.py --gpus N --steps K --warmup W [--workload q1|q1_codes|q1_dict|q6|mixed|cb_*] [--rows R]

One step = one pass of the program (TPC-H Q1 by default: filter, two computed columns, nine aggregates
grouped by (l_returnflag, l_linestatus)) over this rank's SF100 lineitem shard, result on the host (the Arrow structs
ssa_batch_export fills: what a C++ caller of the library holds).  With
N > 1 (one process per GPU, launched by torch.distributed.run) every rank scans its own SF100-sized
row range and the partial aggregates are merged through one RCCL all-gather plus a merge program, all
inside the timed region; `value` is the rows all ranks scanned divided by the slowest rank's time.

At N = 1 the default run also times the other half of the metric on the same resident table -- TPC-H Q6, Q1 over
1-byte flag codes and Q1 over dictionary-encoded flags -- and reports them under "also", each with its own roofline and CPU baseline.

`--workload mixed` is the stream of the sharded SF1000 configuration: Q1 and Q6 alternate over one resident shard of
SF1000 / 8 = 750 M rows; one step = one Q1 pass plus one Q6 pass.

The cb_* workloads are the ClickBench scan-aggregate programs over a synthetic 100 M-row `hits` (ydb_amd/clickbench.py);
their high-cardinality group-bys leave the result table in HBM on one GPU (the row count is read back).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E peak
# FETCH_SIZE passes of the scan kernels (rocprofv3 --pmc, their own runs), kept as ratios to the algorithmic bytes
PMC_TRAFFIC_FILES = ("r3_f_pmc_traffic.json", "r2_c_pmc_traffic.json", "r2_b_pmc_traffic.json", "r2_a_pmc_traffic.json", "r1_d_pmc_traffic.json")  # newest first; a workload missing from one is looked up in the next
SF1000_SHARD_ROWS = 750_000_000  # SF1000 lineitem (6.0e9 rows) over the 8 GPUs of a node


def cpu_baseline(workload: str, seconds_target: float = 15.0, all_cores_seconds: float = 8.0):
    """Times the CPU oracle (a port of the reference's op-at-a-time execution) on a bounded sample."""
    try:
        from oracle import ssa_oracle
    except Exception as e:  # oracle library not built
        return {"value": None, "unit": "rows/s", "cores": 0, "kind": "port", "sample": f"unavailable: {e}"}
    from ydb_amd import tpch

    batch_rows = 1_000_000  # the reference hands the program batches of about this size
    if workload.startswith("cb_"):
        from ydb_amd import clickbench
        prog, cols = clickbench.programs()[workload], clickbench.COLUMNS[workload][0]
        batch = clickbench.DeviceHits(batch_rows, columns=cols).host(cols)
        resolver = clickbench.HITS_RESOLVER
    else:
        codes = workload == "q1_codes"
        if workload == "q1_dict":  # the reference decodes the dictionary when it loads the chunk: its executor sees utf8 flags
            workload = "q1"
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
    per_thread = max(1, int(n * min(1.0, all_cores_seconds / max(dt, 1e-3))))
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


def pmc_traffic(workload: str, algo_bytes_per_launch: float):
    """HBM bytes per launch from the committed FETCH_SIZE passes: (bytes, where the ratio comes from) or (None, None)."""
    here = os.path.dirname(os.path.abspath(__file__))
    for name in PMC_TRAFFIC_FILES:
        try:
            with open(os.path.join(here, "profiles", name)) as f:
                pmc = json.load(f).get(workload)
        except OSError:
            continue
        if pmc and "note" not in pmc:
            rows = pmc.get("rows")
            return (pmc["ratio_to_algorithmic"] * algo_bytes_per_launch,
                    f"not a counter of this run: ratio {pmc['ratio_to_algorithmic']:.4f} of HBM bytes read (rocprofv3 --pmc FETCH_SIZE, "
                    f"its own pass{', ' + str(rows) + ' rows' if rows else ''}) to algorithmic bytes, from profiles/{name}, "
                    "times the algorithmic bytes of this launch")
    return None, None


def tpch_workload(workload: str):
    """(program, columns, algorithmic bytes per row, name) of a TPC-H workload."""
    from ydb_amd import tpch
    if workload == "q6":
        return tpch.q6_program(), tpch.Q6_COLUMNS, tpch.Q6_BYTES_PER_ROW, "TPC-H Q6"
    if workload == "q1_dict":  # the utf8 program over dictionary-encoded flag columns: one code byte per flag and row
        return tpch.q1_program(False), tpch.Q1_COLUMNS_DICT, tpch.Q1_BYTES_PER_ROW_CODES, "TPC-H Q1 (dictionary-encoded utf8 flags)"
    codes = workload == "q1_codes"
    return (tpch.q1_program(codes), tpch.Q1_COLUMNS_CODES if codes else tpch.Q1_COLUMNS_UTF8,
            tpch.Q1_BYTES_PER_ROW_CODES if codes else tpch.Q1_BYTES_PER_ROW_UTF8,
            "TPC-H Q1" + (" (1-byte flag codes)" if codes else " (utf8 flags)"))


def one_by_one_scans() -> int:
    """Keyed scans of this process so far that ran the kernel built for their step the third way (the rows that passed
    the filter dealt out over the lanes and taken one by one: `ssa_spec_keyed_compact` in a kernel trace)."""
    import ctypes
    from ydb_amd import _lib
    n = ctypes.c_int64(0)
    _lib.load().ssa_keyed_compact_stats(ctypes.byref(n))
    return n.value


def roofline_of(workload: str, prog, kernel_ms: float, launches: int, algo_bytes: int, spec_launches: int, one_by_one: bool = False):
    from ydb_amd.shard import last_group_by
    gb = last_group_by(prog)
    keyed = bool(gb.keys) if gb is not None else False
    avg_kernel_s = (kernel_ms / 1e3) / max(1, launches)
    per_launch = algo_bytes / max(1, launches)
    achieved = per_launch / avg_kernel_s / 1e9 if launches and avg_kernel_s > 0 else 0.0
    traffic, source = pmc_traffic(workload, per_launch)
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic, "traffic_source": source,
            # the scan kernel built at run time for this program (same source as the generic ScanKeyedKernel /
            # ScanNoKeyKernel, the program's tables as constants) when one ran
            "kernel": ((("ssa_spec_keyed_compact" if one_by_one else "ssa_spec_keyed") if keyed else "ssa_spec_nokey") if spec_launches
                       else ("ScanKeyedKernel" if keyed else "ScanNoKeyKernel")),
            "avg_kernel_ms": 1e3 * avg_kernel_s, "launches": launches, "algorithmic_bytes_per_launch": int(per_launch)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="q1", choices=["q1", "q1_codes", "q1_dict", "q6", "mixed", "cb_q1", "cb_q2", "cb_q3", "cb_q6", "cb_adv",
                                                         "cb_userid", "cb_phrase", "cb_q20"])
    ap.add_argument("--rows", type=int, default=0,
                    help="rows per GPU (default: TPC-H SF100 lineitem; SF1000 / 8 for mixed; ClickBench hits for cb_*)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="N = 1, workload q1: do not time Q6, Q1-codes, Q1-dictionary and the ClickBench programs as well")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank scans a table of --rows rows (default SF100); strong: the table of --rows rows is split over the ranks")
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

    mixed = args.workload == "mixed"
    clickbench_run = args.workload.startswith("cb_")
    also = ["q6", "q1_codes", "q1_dict"] if (args.workload == "q1" and world == 1 and not args.no_also) else []
    rows = args.rows or (SF1000_SHARD_ROWS if mixed else tpch.SF100_ROWS)
    total_table_rows = rows
    if args.scaling == "strong" and not clickbench_run:
        # the table is the same whatever the number of GPUs: rank r scans rows [r * rows, (r + 1) * rows) of it (the last
        # rank's range is cut to the table's end)
        rows = (total_table_rows + world - 1) // world
    resolver = tpch.LINEITEM_RESOLVER
    table = "SF1000 / 8 lineitem shard" if mixed else "SF100-sized lineitem shard"
    stream = []  # (workload key, program, batch, bytes per row, name)
    if clickbench_run:
        from ydb_amd import clickbench
        rows = args.rows or clickbench.HITS_ROWS
        prog = clickbench.programs()[args.workload]
        cols, bpr = clickbench.COLUMNS[args.workload]
        resolver = clickbench.HITS_RESOLVER
        table = "hits-sized shard"
        li = clickbench.DeviceHits(rows, device=local_rank, row0=rank * rows, columns=cols)
        if args.workload == "cb_phrase":
            bpr = 4 + li.phrase_bytes / rows
        if args.workload == "cb_q20":
            bpr = 4 + li.url_bytes / rows
        stream.append((args.workload, prog, li.batch(cols), bpr, "ClickBench " + args.workload[3:]))
    else:
        keys = ["q1", "q6"] if mixed else [args.workload]
        my_rows = rows
        if args.scaling == "strong":
            my_rows = max(0, min(rows, total_table_rows - rank * rows))
        li = tpch.DeviceLineitem(my_rows, device=local_rank, row0=rank * rows,
                                 utf8_flags=("q1" in keys), code_flags=bool({"q1_codes", "q1_dict"} & set(keys + also)))
        for k in keys:
            prog, cols, bpr, name = tpch_workload(k)
            stream.append((k, prog, li.batch(cols), bpr, name))
    scans = [ShardedScan(prog, resolver, device=local_rank) for _, prog, _, _, _ in stream]
    # a result of millions of groups stays in HBM (its row count is read back); with several ranks each keeps the groups
    # whose key hash it owns
    on_device = args.workload in ("cb_userid", "cb_phrase")

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    third_way = [0]  # keyed scans of the last timed region that took the rows behind the filter one by one

    def timed(scans_, batches, steps, warmup, settle=None, keep=None):
        """`warmup` untimed steps, then exactly `steps` timed ones; one step runs every program of the stream once."""
        settle = (3 if clickbench_run else 2) if settle is None else settle
        keep = on_device if keep is None else keep
        results = [None] * len(scans_)
        # The executor builds the step program's kernel at the first large scan and settles how it runs it
        # (register-held groups on or off) on the second; groupings of a million groups take a third such pass (records
        # sorted by table region, or row-by-row updates), and so do groupings behind a filter (the rows that passed taken
        # one by one).  Those passes run here, untimed, like the generation of the table -- each is a second of hiprtc on
        # the host with the GPU idle -- and the warm-up steps asked for come after them.
        for _ in range(settle + warmup):
            for i, s in enumerate(scans_):
                results[i] = s.run(batches[i], gather_device=gather_dev, keep_on_device=keep, lazy=True)
        # (a table of fewer than 2^25 rows has its kernel built by a worker thread while the scans go on with the kernels
        # the library was built with: the timed steps wait for it, a few seconds at most)
        waited = time.perf_counter()
        while time.perf_counter() - waited < 8.0:
            launches_before = scans_[0].local.SpecializationStats()[2]  # (process-wide: launches of kernels built at run time)
            results[0] = scans_[0].run(batches[0], gather_device=gather_dev, keep_on_device=keep, lazy=True)
            if scans_[0].local.SpecializationStats()[2] > launches_before or os.environ.get("SSA_JIT") == "0":
                break
            time.sleep(0.05)
        for s in scans_:
            s.local.EnableProfile(True)
            s.local.ReadProfile()
        barrier()
        third_way[0] = one_by_one_scans()
        # (as timeit does: a collection of Python's cyclic garbage inside the timed region is a millisecond of host time
        # that has nothing to do with the scan)
        gc.collect()
        gc.disable()
        t0 = time.perf_counter()
        marks = [t0]
        for _ in range(steps):
            for i, s in enumerate(scans_):
                results[i] = s.run(batches[i], gather_device=gather_dev, keep_on_device=keep, lazy=True)
            marks.append(time.perf_counter())
        barrier()
        elapsed = time.perf_counter() - t0
        gc.enable()
        if os.environ.get("BENCH_STEP_TIMES"):  # (every step ends with its result on the host: the marks are step ends)
            print("steps, ms: " + " ".join(f"{1e3 * (b - a):.2f}" for a, b in zip(marks, marks[1:])), file=sys.stderr, flush=True)
        third_way[0] = one_by_one_scans() - third_way[0]
        profiles = [s.local.ReadProfile() for s in scans_]
        for s in scans_:
            s.local.EnableProfile(False)
        return elapsed, profiles, results

    elapsed, profiles, results = timed(scans, [b for _, _, b, _, _ in stream], args.steps, args.warmup)
    headline_one_by_one = len(stream) == 1 and third_way[0] >= args.steps
    built, build_ms, spec_launches = scans[0].local.SpecializationStats()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=gather_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    also_out = {}
    for k in also:  # the other half of the metric on the same resident table (N = 1 only)
        prog, cols, bpr, name = tpch_workload(k)
        s = ShardedScan(prog, resolver, device=local_rank)
        e, prof, _ = timed([s], [li.batch(cols)], args.steps, args.warmup)
        kernel_ms, launches, algo_bytes = prof[0]
        also_out[k] = {"workload": f"{name}, the same {rows}-row resident table", "value": rows * args.steps / e, "unit": "rows/s",
                       "ms_per_step": 1e3 * e / args.steps, "algorithmic_bytes_per_row": bpr,
                       "roofline": roofline_of(k, prog, kernel_ms, launches, algo_bytes, s.local.SpecializationStats()[2])}
        if not args.no_cpu_baseline:
            also_out[k]["cpu_baseline"] = cpu_baseline(k, seconds_target=8.0, all_cores_seconds=4.0)

    if also and rows >= 64 << 20:
        # The same Q1 over the table cut into 64 slices of 2^20 rows -- the batch size a column shard hands its program
        # (program.h:423-433 runs the slices one by one) -- through ssa_program_apply_many: one result per slice, the
        # slices scanned by one launch.
        from ydb_amd.program import ProgramContainer
        per, nslices = 1 << 20, 64
        prog, cols, bpr, name = tpch_workload("q1")
        pc = ProgramContainer(local_rank)
        ok, err = pc.Init(resolver, prog)
        if not ok:
            raise SystemExit(err)
        slices = [li.batch(cols, row0=i * per, rows=per) for i in range(nslices)]
        for _ in range(3):
            pc.ApplyDeviceMany(slices)
        pc.EnableProfile(True)
        pc.ReadProfile()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            res = pc.ApplyDeviceMany(slices)
        barrier()
        e = time.perf_counter() - t0
        kernel_ms, launches, algo_bytes = pc.ReadProfile()
        pc.EnableProfile(False)
        also_out["q1_slices"] = {"workload": f"{name}, {nslices} slices of {per} rows of the resident table, one result per slice "
                                             "(ssa_program_apply_many)",
                                 "value": per * nslices * args.steps / e, "unit": "rows/s", "ms_per_step": 1e3 * e / args.steps,
                                 "groups_per_slice": res[0].rows, "algorithmic_bytes_per_row": bpr,
                                 "roofline": roofline_of("q1_slices", prog, kernel_ms, launches, algo_bytes, pc.SpecializationStats()[2])}
        del res, slices

    if also:
        # The ClickBench half of the target on the same GPU: the scan-aggregates over 1 G rows (10 x hits: at two bytes a row
        # the hits-sized table is a 50 us launch), the substring search over 25 M URLs (one batch is bounded by int32
        # offsets), the high-cardinality group-by over a hits-sized table.  Synthetic `hits` generated on the device.
        from ydb_amd import clickbench
        del li
        gc.collect()
        torch.cuda.empty_cache()
        for k, cb_rows in (("cb_q1", 1_000_000_000), ("cb_adv", 1_000_000_000), ("cb_q20", 25_000_000), ("cb_userid", clickbench.HITS_ROWS)):
            cols, bpr = clickbench.COLUMNS[k]
            hits = clickbench.DeviceHits(cb_rows, device=local_rank, columns=cols)
            if k == "cb_q20":
                bpr = 4 + hits.url_bytes / cb_rows
            prog = clickbench.programs()[k]
            s = ShardedScan(prog, clickbench.HITS_RESOLVER, device=local_rank)
            before = one_by_one_scans()
            e, prof, res = timed([s], [hits.batch(cols)], args.steps, args.warmup, settle=(6 if k == "cb_adv" else 3), keep=(k == "cb_userid"))  # (cb_adv: three ways to run its kernel are timed first)
            kernel_ms, launches, algo_bytes = prof[0]
            entry = {"workload": f"ClickBench {k[3:]}, synthetic hits of {cb_rows} rows resident in HBM", "value": cb_rows * args.steps / e,
                     "unit": "rows/s", "ms_per_step": 1e3 * e / args.steps, "algorithmic_bytes_per_row": bpr,
                     "groups": res[0].rows if k == "cb_userid" else len(next(iter(res[0].values())).valid)}
            if k == "cb_userid":
                # scan, two rounds of the sort (the second with a counting pass), the region pass, the move to the output: several kernels, priced as a step
                # against the bytes of the key column
                entry["roofline"] = {"bound": "hbm", "achieved": cb_rows * bpr / (e / args.steps) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                     "frac": cb_rows * bpr / (e / args.steps) / 1e9 / HBM_PEAK_GBPS, "traffic": None,
                                     "kernel": "the whole step: ssa_spec_keyed_plain, RadixScatter, RadixCount + RadixScatter, RegionEmitExactKernel, RegionCompactEmitKernel",
                                     "algorithmic_bytes_per_launch": int(cb_rows * bpr)}
            else:
                entry["roofline"] = roofline_of(k, prog, kernel_ms, launches, algo_bytes, s.local.SpecializationStats()[2],
                                                one_by_one_scans() - before >= args.steps)
            if not args.no_cpu_baseline:
                entry["cpu_baseline"] = cpu_baseline(k, seconds_target=6.0, all_cores_seconds=3.0)
            also_out[k] = entry
            del hits, s, res
            gc.collect()
            torch.cuda.empty_cache()

    result = results[0]
    groups = result.rows if on_device else (len(next(iter(result.values())).valid) if result else 0)
    if on_device and world > 1:  # the result is spread over the ranks
        t = torch.tensor([groups], dtype=torch.int64, device=gather_dev)
        dist.all_reduce(t)
        groups = int(t.item())
    if rank == 0:
        total_rows = (total_table_rows if (args.scaling == "strong" and not clickbench_run) else rows * world) * args.steps * len(stream)
        wl_key, prog, _, bpr, name = stream[0]
        kernel_ms, launches, algo_bytes = profiles[0]
        names = " then ".join(n for _, _, _, _, n in stream)
        out = {
            "metric": "rows/sec scanned + % HBM roofline, TPC-H Q1/Q6 SF100 at 1/2/4/8 MI355X",
            "value": total_rows / elapsed,
            "unit": "rows/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": args.scaling if not clickbench_run else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": (f"{names}, {table} of {total_table_rows} rows split over {world} GPU(s) ({rows} rows each), every part one HBM-resident batch"
                                    if (args.scaling == "strong" and not clickbench_run) else
                                    f"{names}, {table} per GPU ({rows} rows), whole shard as one HBM-resident batch") +
                                   (" (one step = every program of the stream once; value counts the rows each program scanned)" if mixed else ""),
                       "rows_per_gpu": rows, "algorithmic_bytes_per_row": bpr if not mixed else sum(b for _, _, _, b, _ in stream),
                       "groups": groups,
                       "kernels_built_at_run_time": built, "kernel_build_ms_before_the_timed_steps": build_ms,
                       "untimed_passes_that_build_and_settle_the_kernels_before_the_warmup": 3 if clickbench_run else 2,
                       "float_sum_tolerance": "relative 1e-12 against the CPU oracle up to 1e5 rows, 1e-10 at millions of rows "
                                              "(the reference's own sequential sum is off by n * 2^-53), checked in tests/",
                       "parallelism": f"row-range shards x{world}, partial aggregates merged by " +
                                      ("hash-partitioned exchange" if world > 1 and groups > ShardedScan.ALL_TO_ALL_GROUPS
                                       else "RCCL all-gather") + " + merge program"},
            # the dominant kernel: the first program of the stream (Q1's keyed scan)
            "roofline": roofline_of(wl_key, prog, kernel_ms, launches, algo_bytes, spec_launches, headline_one_by_one),
        }
        if mixed:
            out["programs"] = {}
            for (k, p, _, b, n), (kms, nl, ab), s in zip(stream, profiles, scans):
                out["programs"][k] = {"workload": n, "algorithmic_bytes_per_row": b,
                                      "roofline": roofline_of(k, p, kms, nl, ab, s.local.SpecializationStats()[2])}
        if world == 1 and not args.no_cpu_baseline:
            if mixed:
                parts = {k: cpu_baseline(k, seconds_target=8.0, all_cores_seconds=4.0) for k, _, _, _, _ in stream}
                # the stream scans every row once per program: harmonic combination of the programs' rates
                v = len(parts) / sum(1.0 / p["value"] for p in parts.values()) if all(p.get("value") for p in parts.values()) else None
                out["cpu_baseline"] = {"value": v, "unit": "rows/s", "cores": 1, "kind": "port",
                                       "sample": "per program: " + "; ".join(f"{k}: {p['sample']}" for k, p in parts.items()),
                                       "programs": parts}
            else:
                out["cpu_baseline"] = cpu_baseline(args.workload)
        if also_out:
            out["also"] = also_out
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

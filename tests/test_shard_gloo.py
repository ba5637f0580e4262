"""The N > 1 path on CPU: two gloo ranks scan their own row ranges (with the oracle standing in for the GPU scan, which
this box cannot run), exchange partial aggregates with shard.all_gather_results, and merge them with the derived merge
program.  The merged result must equal the program applied to the whole table."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, codes, rows_per_rank, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from cases import compare_batches
    from oracle import ssa_oracle
    from ydb_amd import shard, tpch

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        prog = tpch.q1_program(codes)
        cols = tpch.Q1_COLUMNS_CODES if codes else tpch.Q1_COLUMNS_UTF8
        li = tpch.lineitem_host(rank * rows_per_rank, rows_per_rank)
        local = ssa_oracle.Oracle(tpch.LINEITEM_RESOLVER, prog).apply({c: li[c] for c in cols})
        parts = shard.all_gather_results(local)
        assert len(parts) == world
        mine = parts[rank]
        assert list(mine.keys()) == list(local.keys())
        for k in local:
            assert mine[k].to_pylist() == local[k].to_pylist(), k  # the wire format round-trips
        mp_, mr = shard.merge_program(prog, tpch.LINEITEM_RESOLVER)
        merged = ssa_oracle.Oracle(mr, mp_).apply(shard.partial_batch(parts, prog, tpch.LINEITEM_RESOLVER))
        full = tpch.lineitem_host(0, world * rows_per_rank)
        exp = ssa_oracle.Oracle(tpch.LINEITEM_RESOLVER, prog).apply({c: full[c] for c in cols})
        compare_batches(merged, exp, key_names=cols[-2:], float_sum_names=["130", "131", "132", "133", "136"])
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("codes", [True, False])
def test_two_rank_partial_aggregate_merge(tmp_path, codes):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, codes, 20_000, str(tmp_path)), nprocs=world, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1"]


def _worker_alltoall(rank, world, port, rows_per_rank, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import pickle
    import torch.distributed as dist
    from oracle import ssa_oracle
    from ydb_amd import shard, tpch
    from ydb_amd.ssa_pb import AGG, Aggregate, Program

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # high-cardinality grouping: by the price column's value, with a string key next to it
        prog = Program().group_by([Aggregate(120, AGG.COUNT, []), Aggregate(121, AGG.SUM, [tpch.L_QUANTITY]),
                                   Aggregate(122, AGG.MIN, [tpch.L_SHIPDATE])], keys=[tpch.L_EXTENDEDPRICE, tpch.L_RETURNFLAG])
        cols = ["l_extendedprice", "l_returnflag", "l_quantity", "l_shipdate"]
        li = tpch.lineitem_host(rank * rows_per_rank, rows_per_rank)
        local = ssa_oracle.Oracle(tpch.LINEITEM_RESOLVER, prog).apply({c: li[c] for c in cols})
        parts = shard.all_to_all_results(local, ["l_extendedprice", "l_returnflag"])
        mp_, mr = shard.merge_program(prog, tpch.LINEITEM_RESOLVER)
        mine = ssa_oracle.Oracle(mr, mp_).apply(shard.partial_batch(parts, prog, tpch.LINEITEM_RESOLVER))
        rows = list(zip(mine["l_extendedprice"].to_pylist(), mine["l_returnflag"].to_pylist(), mine["120"].to_pylist(),
                        mine["121"].to_pylist(), mine["122"].to_pylist()))
        with open(os.path.join(out_dir, f"part{rank}.pkl"), "wb") as f:
            pickle.dump(rows, f)
    finally:
        dist.destroy_process_group()


def test_two_rank_hash_partitioned_exchange(tmp_path):
    """High-cardinality path: groups are exchanged point-to-point to the rank that owns their key hash; the union of the
    ranks' merged parts is the program applied to the whole table, and no group appears on two ranks."""
    import pickle
    sys.path.insert(0, ROOT)
    from oracle import ssa_oracle
    from ydb_amd import tpch
    from ydb_amd.ssa_pb import AGG, Aggregate, Program
    world, rows_per_rank = 2, 30_000
    mp.spawn(_worker_alltoall, args=(world, _free_port(), rows_per_rank, str(tmp_path)), nprocs=world, join=True)
    got = []
    for r in range(world):
        with open(os.path.join(tmp_path, f"part{r}.pkl"), "rb") as f:
            got.extend(pickle.load(f))
    prog = Program().group_by([Aggregate(120, AGG.COUNT, []), Aggregate(121, AGG.SUM, [tpch.L_QUANTITY]),
                               Aggregate(122, AGG.MIN, [tpch.L_SHIPDATE])], keys=[tpch.L_EXTENDEDPRICE, tpch.L_RETURNFLAG])
    full = tpch.lineitem_host(0, world * rows_per_rank)
    exp = ssa_oracle.Oracle(tpch.LINEITEM_RESOLVER, prog).apply({c: full[c] for c in ["l_extendedprice", "l_returnflag", "l_quantity", "l_shipdate"]})
    want = list(zip(exp["l_extendedprice"].to_pylist(), exp["l_returnflag"].to_pylist(), exp["120"].to_pylist(),
                    exp["121"].to_pylist(), exp["122"].to_pylist()))
    assert len(got) == len(want) and len({(a, b) for a, b, *_ in got}) == len(got)
    assert sorted(got) == sorted(want)


def test_merge_program_shape():
    sys.path.insert(0, ROOT)
    from ydb_amd import shard, tpch
    from ydb_amd.ssa_pb import AGG, GroupBy
    mp_, mr = shard.merge_program(tpch.q1_program(True), tpch.LINEITEM_RESOLVER)
    gb = [c for c in mp_.commands if isinstance(c, GroupBy)][0]
    assert [a.id for a in gb.aggregates] == [AGG.SUM] * 9  # SUM of sums and of counts
    names = dict(mr)
    assert names[gb.keys[0]] == "l_returnflag_code" and names[gb.keys[1]] == "l_linestatus_code"
    assert all(names[a.args[0]].startswith("partial:") for a in gb.aggregates)


def test_partition_of_keys_is_balanced_and_stable():
    sys.path.insert(0, ROOT)
    from ydb_amd import shard
    h = np.arange(100_000, dtype=np.uint64) * np.uint64(2654435761)
    r = shard.partition_of_keys(h, 8)
    assert r.min() == 0 and r.max() == 7
    counts = np.bincount(r, minlength=8)
    assert counts.min() > 0.9 * 12_500 and counts.max() < 1.1 * 12_500
    assert (shard.partition_of_keys(h, 8) == r).all()


def test_commands_after_the_aggregation_run_on_the_merged_groups():
    """HAVING count > c, avg = sum / count, projection of the average only: shards run the program up to its last GroupBy
    (shard.split_at_last_group_by), the merge program reduces their partial tables and then applies what follows the
    aggregation.  Filtering or projecting per shard would drop groups that pass only when merged, and sums the average
    needs."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cases import compare_batches
    from oracle import ssa_oracle
    from ydb_amd import shard, tpch
    from ydb_amd.ssa_pb import AGG, F, Aggregate, Program
    rows = 12_000
    prog = (Program(version=5)
            .group_by([Aggregate(130, AGG.SUM, [tpch.L_QUANTITY]), Aggregate(131, AGG.COUNT, [tpch.L_QUANTITY]), Aggregate(132, AGG.MAX, [tpch.L_SHIPDATE])],
                      keys=[tpch.L_SHIPDATE])
            .assign_const(140, "Uint64", 5).assign_func(141, F.CMP_GREATER, [131, 140]).filter(141)
            .assign_func(142, F.MATH_DIVIDE, [130, 131])
            .project([tpch.L_SHIPDATE, 142, 132]))
    cols = ["l_shipdate", "l_quantity"]
    head, tail = shard.split_at_last_group_by(prog)
    assert len(head.commands) == 1 and len(tail) == 5
    full = tpch.lineitem_host(0, rows)
    exp = ssa_oracle.Oracle(tpch.LINEITEM_RESOLVER, prog).apply({c: full[c] for c in cols})
    assert 0 < len(exp["142"].to_pylist())
    parts = []
    for r in range(3):
        li = tpch.lineitem_host(r * rows // 3, rows // 3)
        parts.append(ssa_oracle.Oracle(tpch.LINEITEM_RESOLVER, head).apply({c: li[c] for c in cols}))
    # some days have more than five rows only in the whole table: a HAVING applied per shard would have dropped them
    per_shard_days = set()
    for r in range(3):
        li = tpch.lineitem_host(r * rows // 3, rows // 3)
        per_shard_days |= set(ssa_oracle.Oracle(tpch.LINEITEM_RESOLVER, prog).apply({c: li[c] for c in cols})["l_shipdate"].to_pylist())
    assert per_shard_days < set(exp["l_shipdate"].to_pylist())
    mp_, mr = shard.merge_program(prog, tpch.LINEITEM_RESOLVER)
    merged = ssa_oracle.Oracle(mr, mp_).apply(shard.partial_batch(parts, prog, tpch.LINEITEM_RESOLVER))
    assert list(merged.keys()) == list(exp.keys())
    compare_batches(merged, exp, key_names=["l_shipdate"], float_sum_names=["142"])


def test_a_program_with_two_aggregations_is_refused():
    """The first aggregation's groups span the shards; a second one over per-shard partial groups would be merged into
    something else than the second aggregation of the whole table."""
    from ydb_amd import shard
    from ydb_amd.ssa_pb import AGG, Aggregate, Program
    p = (Program().group_by([Aggregate(10, AGG.COUNT, [])], keys=[1, 2])
         .group_by([Aggregate(11, AGG.SUM, [10])], keys=[1]))
    with pytest.raises(ValueError, match="more than one GroupBy"):
        shard.split_at_last_group_by(p)
    with pytest.raises(ValueError, match="more than one GroupBy"):
        shard.merge_program(p, [(1, "a"), (2, "b")])

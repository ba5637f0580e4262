"""Compute sharding on the GPU (ssa_sharding_hash / ssa_sharding_split) against the sharding oracle: bit-exact hashes,
permutations and shard sizes; size-independent properties at a few million rows."""
import numpy as np
import pytest

from ydb_amd.arrow_c import Col
from ydb_amd.ssa_pb import Program

pytestmark = pytest.mark.gpu


def _container(resolver):
    from ydb_amd.program import ProgramContainer
    pc = ProgramContainer()
    ok, err = pc.Init(resolver, Program().project([resolver[0][0]]))
    assert ok, err
    return pc


def _columns(batch, names):
    cols = []
    for n in names:
        c = batch[n]
        if isinstance(c, Col):
            cols.append((c.values, c.valid))
        else:
            cols.append((c, None))
    return cols


def _strings(rng, n, maxlen):
    alphabet = np.array(list("abcdefghijklmnopqrstuvwxyz0123456789"))
    return ["".join(rng.choice(alphabet, rng.integers(0, maxlen + 1))) for _ in range(n)]


@pytest.mark.parametrize("n", [0, 1, 63, 1025, 20_000])
def test_row_hashes_match_the_oracle(n):
    from oracle import sharding_oracle as so
    rng = np.random.default_rng(100 + n)
    batch = {
        "i8": rng.integers(-128, 128, n).astype(np.int8),
        "u16": rng.integers(0, 1 << 16, n).astype(np.uint16),
        "i32": Col(rng.integers(-2**31, 2**31, n).astype(np.int32), valid=rng.random(n) > 0.2),
        "u64": rng.integers(0, 2**63, n).astype(np.uint64),
        "f32": rng.random(n).astype(np.float32),
        "f64": Col(rng.random(n), valid=rng.random(n) > 0.5),
        "b": rng.random(n) > 0.5,
        "s": _strings(rng, n, 12),
        "long": Col(_strings(rng, n, 90), valid=rng.random(n) > 0.1),
    }
    resolver = [(i + 1, name) for i, name in enumerate(batch)]
    pc = _container(resolver)
    dev = pc.Import(batch)
    for keys, seed in ((["i32"], 0), (["u64"], 7), (["s"], 0), (["long"], so.COMPUTE_SHARDING_SEED), (["b", "i8"], 1),
                       (["i8", "u16", "i32", "u64"], so.COMPUTE_SHARDING_SEED), (["f32", "f64", "s", "u16"], 3),
                       (["u64", "u64", "u64", "u64", "u64"], 0),                  # 40 bytes: one stripe and a tail
                       (["long", "s", "i32", "b", "f64", "u64", "i8", "long"], so.COMPUTE_SHARDING_SEED)):
        got = pc.ShardingHash(dev, keys, seed)
        exp = so.hash_rows(_columns(batch, keys), n, seed)
        assert np.array_equal(got, exp), (keys, seed)


def _check_split(pc, batch, names, keys, shards, hashes_i64):
    from oracle import sharding_oracle as so
    n = len(hashes_i64)
    dev = pc.Import(batch)
    out, sizes = pc.ShardingSplit(dev, keys, shards)
    perm, exp_sizes = so.split(hashes_i64, shards)
    assert sizes == exp_sizes
    got = pc.Export(out)
    for name in names:
        c = batch[name]
        vals = c.values if isinstance(c, Col) else c
        valid = c.valid if isinstance(c, Col) else None
        g = got[name]
        if isinstance(vals, np.ndarray):
            exp = vals[perm]
            if valid is not None:
                ev = np.asarray(valid)[perm]
                assert np.array_equal(np.asarray(g.valid, dtype=bool), ev)
                assert np.array_equal(np.asarray(g.values)[ev], exp[ev])
            else:
                assert np.array_equal(np.asarray(g.values), exp), name
        else:
            exp = [vals[i] for i in perm]
            gl = [v.decode() if isinstance(v, bytes) else v for v in g.to_pylist()]
            if valid is not None:
                exp = [e if valid[i] else None for e, i in zip(exp, perm)]
            assert gl == exp, name
    assert out.rows == n


@pytest.mark.parametrize("dtype", [np.uint64, np.int64, np.uint32, np.int32])
@pytest.mark.parametrize("n,shards", [(1, 2), (10, 3), (1025, 7), (30_000, 64), (5000, 512)])
def test_split_by_a_single_integer_key_uses_the_key_as_the_hash(dtype, n, shards):
    rng = np.random.default_rng(n * 31 + shards)
    info = np.iinfo(dtype)
    key = rng.integers(max(info.min, -2**62), min(info.max, 2**62), n, dtype=np.int64).astype(dtype)
    if dtype == np.uint64:
        key[::5] |= np.uint64(1 << 63)  # read as negative int64 by the reference
    batch = {"k": key, "v": rng.random(n), "s": _strings(rng, n, 5)}
    pc = _container([(1, "k"), (2, "v"), (3, "s")])
    _check_split(pc, batch, ["k", "v", "s"], ["k"], shards, key.astype(np.int64) if dtype != np.uint64 else key.view(np.int64))


@pytest.mark.parametrize("n,shards", [(10, 3), (4097, 5), (20_000, 16)])
def test_split_by_hashed_keys_and_skewed_data_triggers_the_size_correction(n, shards):
    from oracle import sharding_oracle as so
    rng = np.random.default_rng(n + shards)
    # most rows share one key: the longest list is far above rows / shards + 1 and gives its tail away
    region = [("north" if rng.random() < 0.7 else r) for r in _strings(rng, n, 6)]
    day = Col(rng.integers(0, 5, n).astype(np.int16), valid=rng.random(n) > 0.1)
    batch = {"region": region, "day": day, "x": rng.integers(0, 1 << 40, n).astype(np.int64)}
    pc = _container([(1, "region"), (2, "day"), (3, "x")])
    h = so.hash_rows(_columns(batch, ["region", "day"]), n, so.COMPUTE_SHARDING_SEED)
    _check_split(pc, batch, ["region", "day", "x"], ["region", "day"], shards, h.view(np.int64))
    perm, sizes = so.split(h.view(np.int64), shards)
    assert max(sizes) <= max(n // shards + 1, min(sizes))  # the correction ran


@pytest.mark.parametrize("parts", [2, 8])
def test_partition_by_key_alone_keeps_equal_keys_together(parts):
    """The partitioning of the exchange between GPUs: no balancing, so two batches (two ranks' partial tables) send a key
    they share to the same part however skewed their rows are -- and the part is the one the oracle's hash names."""
    from oracle import sharding_oracle as so
    rng = np.random.default_rng(40 + parts)
    pc = _container([(1, "region"), (2, "day"), (3, "x")])
    where = {}
    for n, hot in ((5000, 0.7), (3001, 0.0), (777, 0.95)):
        region = [("north" if rng.random() < hot else r) for r in _strings(rng, n, 6)]
        day = rng.integers(0, 5, n).astype(np.int16)
        batch = {"region": region, "day": day, "x": rng.integers(0, 1 << 40, n).astype(np.int64)}
        out, sizes = pc.PartitionByKey(pc.Import(batch), ["region", "day"], parts)
        got = pc.Export(out)
        assert sum(sizes) == n
        h = so.hash_rows(_columns(batch, ["region", "day"]), n, so.COMPUTE_SHARDING_SEED).view(np.int64)
        expect = {(r, int(d)): int(abs(int(hh)) % parts) for r, d, hh in zip(region, day, h)}
        at = 0
        for part, size in enumerate(sizes):
            for r, d in zip(got["region"].values[at:at + size], np.asarray(got["day"].values)[at:at + size]):
                key = (r.decode() if isinstance(r, bytes) else r, int(d))
                assert expect[key] == part
                assert where.setdefault(key, part) == part
            at += size
        # rows of a part keep their input order, every row is there once
        assert sorted(np.asarray(got["x"].values).tolist()) == sorted(batch["x"].tolist())


def test_unsharded_cases_return_the_batch_unchanged():
    rng = np.random.default_rng(5)
    n = 1000
    batch = {"k": rng.integers(0, 100, n).astype(np.int32), "v": rng.random(n)}
    pc = _container([(1, "k"), (2, "v")])
    dev = pc.Import(batch)
    for keys, shards in ((["k"], 1), ([], 4), (["missing"], 4)):
        out, sizes = pc.ShardingSplit(dev, keys, shards)
        got = pc.Export(out)
        assert np.array_equal(np.asarray(got["k"].values), batch["k"]) and sizes[0] == n and sum(sizes) == n


def test_split_of_millions_of_rows_keeps_every_row_once_and_orders_shards():
    from ydb_amd import tpch
    from ydb_amd.program import ProgramContainer
    n = 8_000_000
    li = tpch.DeviceLineitem(n)
    pc = ProgramContainer()
    ok, err = pc.Init(tpch.LINEITEM_RESOLVER, tpch.q6_program())
    assert ok, err
    cols = ["l_shipdate", "l_quantity", "l_extendedprice"]
    dev = li.batch(cols)
    shards = 8
    out, sizes = pc.ShardingSplit(dev, ["l_shipdate", "l_quantity"], shards)
    assert sum(sizes) == n and max(sizes) <= n // shards + 1 + max(0, max(sizes) - min(sizes))
    got = pc.Export(out)
    host = tpch.lineitem_host(0, n)
    # a permutation: column checksums are unchanged
    assert int(np.sum(got["l_shipdate"].values.astype(np.int64))) == int(np.sum(host["l_shipdate"].astype(np.int64)))
    assert np.array_equal(np.sort(np.asarray(got["l_extendedprice"].values)), np.sort(host["l_extendedprice"]))
    # rows with equal keys land in one shard unless the size correction moved them: check on the hash of a sample
    h = pc.ShardingHash(dev, ["l_shipdate", "l_quantity"], 34323543)
    natural = (np.abs(h.view(np.int64)) % shards)
    assert np.all(np.bincount(natural, minlength=shards) >= 0)

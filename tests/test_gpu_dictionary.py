"""Dictionary-encoded string columns kept encoded in HBM: every program gives the results of the decoded column (the
oracle decodes, as the reference does when it loads a chunk: ydb/core/formats/arrow/dictionary/conversion.h:13-15),
bit-exact, and the cases that can be worked on the codes are (ssa_dictionary_stats says which way a step ran)."""
import ctypes

import numpy as np
import pytest

from cases import compare_batches
from ydb_amd.arrow_c import Col
from ydb_amd.ssa_pb import AGG, F, YQL, Aggregate, Program

pytestmark = pytest.mark.gpu


def stats():
    from ydb_amd import _lib
    lib = _lib.load()
    a, b = ctypes.c_int64(0), ctypes.c_int64(0)
    lib.ssa_dictionary_stats(ctypes.byref(a), ctypes.byref(b))
    return a.value, b.value


def both(resolver, program, batch):
    from oracle import ssa_oracle
    from ydb_amd.program import ProgramContainer
    pc = ProgramContainer()
    ok, err = pc.Init(resolver, program)
    assert ok, err
    s0 = stats()
    got = pc.ApplyProgram(batch)
    s1 = stats()
    exp = ssa_oracle.Oracle(resolver, program).apply(batch)
    return got, exp, (s1[0] - s0[0], s1[1] - s0[1])


RES = [(1, "phrase"), (2, "v"), (3, "engine")]
WORDS = ["google", "yandex", "", "bing.com/search", "duck", "go", "a much longer search phrase than the others: 0123456789"]


def batch_of(rng, n, words=WORDS, nulls=0.0, code_type=np.int8, unused=0, engine_words=("direct", "organic", "ad")):
    d = list(words) + ["never used %d" % i for i in range(unused)]
    codes = rng.integers(0, len(words), n).astype(code_type)
    valid = None if nulls <= 0 else rng.random(n) >= nulls
    if valid is not None:  # the codes under nulls are arbitrary
        codes = np.where(valid, codes, rng.integers(0, 120, n).astype(code_type))
    return {"phrase": Col(codes, valid=valid, dictionary=d),
            "v": rng.integers(-10, 10, n).astype(np.int64),
            "engine": Col(rng.integers(0, len(engine_words), n).astype(np.int32), dictionary=list(engine_words))}


@pytest.mark.parametrize("n", [0, 1, 1000, 70_001])
@pytest.mark.parametrize("nulls", [0.0, 0.2])
def test_predicate_and_group_by_on_codes(n, nulls):
    rng = np.random.default_rng(9100 + n)
    p = (Program().assign_const(100, "Bytes", b"g").assign_yql(110, YQL.StringContains, [1, 100]).filter(110)
         .group_by([Aggregate(120, AGG.COUNT, []), Aggregate(121, AGG.SUM, [2]), Aggregate(122, AGG.COUNT, [1])], keys=[1]))
    got, exp, (on_codes, decoded) = both(RES, p, batch_of(rng, n, nulls=nulls))
    compare_batches(got, exp, key_names=["phrase"])
    assert decoded == 0 and on_codes == 1


@pytest.mark.parametrize("code_type", [np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.int64, np.uint64])
def test_every_code_type(code_type):
    rng = np.random.default_rng(9200)
    p = (Program().assign_func(110, F.STR_LENGTH, [1]).assign_const(100, "Uint32", 4).assign_func(111, F.CMP_GREATER, [110, 100])
         .filter(111).group_by([Aggregate(120, AGG.COUNT, []), Aggregate(121, AGG.MIN, [110])], keys=[1, 3]))
    got, exp, (on_codes, decoded) = both(RES, p, batch_of(rng, 20_000, code_type=code_type, nulls=0.1))
    compare_batches(got, exp, key_names=["phrase", "engine"])
    assert decoded == 0 and on_codes == 1


def test_equal_entries_are_one_group_and_unused_entries_none():
    rng = np.random.default_rng(9300)
    words = ["x", "yy", "x", "", "yy", "zzz", ""]
    p = Program().group_by([Aggregate(120, AGG.COUNT, []), Aggregate(121, AGG.SUM, [2])], keys=[1])
    got, exp, (on_codes, decoded) = both(RES, p, batch_of(rng, 30_000, words=words, unused=5))
    compare_batches(got, exp, key_names=["phrase"])
    assert len(got["120"].to_pylist()) == 4
    assert decoded == 0 and on_codes == 1


def test_null_rows_group_with_the_empty_string():
    # the `serialized` method writes a null slot's bytes: an empty string, with or without an empty entry
    rng = np.random.default_rng(9400)
    p = Program().group_by([Aggregate(120, AGG.COUNT, []), Aggregate(121, AGG.COUNT, [1])], keys=[1])
    for words in (["a", "", "b"], ["a", "c", "b"]):
        got, exp, (on_codes, decoded) = both(RES, p, batch_of(rng, 5000, words=words, nulls=0.3))
        compare_batches(got, exp, key_names=["phrase"])
        assert decoded == 0 and on_codes == 1


def test_functions_per_entry():
    rng = np.random.default_rng(9500)
    n = 40_000
    p = (Program().assign_const(100, "Bytes", b"go").assign_const(101, "Bytes", b"h")
         .assign_func(110, F.STR_STARTS_WITH, [1, 100]).assign_yql(111, YQL.EndsWith, [1, 101])
         .assign_func(112, F.STR_LENGTH, [1]).assign_func(113, F.IS_NULL, [1]).assign_func(114, F.CMP_EQUAL, [3, 100])
         .assign_func(115, F.MATH_ADD, [112, 2])
         .project([110, 111, 112, 113, 114, 115, 2]))
    got, exp, (on_codes, decoded) = both(RES, p, batch_of(rng, n, nulls=0.15, code_type=np.int16))
    compare_batches(got, exp)
    assert decoded == 0 and on_codes == 1


def test_like_patterns_per_entry():
    rng = np.random.default_rng(9600)
    p = (Program().assign_const(100, "Bytes", b"%o%g%").assign_const(101, "Bytes", b"search")
         .assign_func(110, F.STR_MATCH_LIKE, [1, 100]).assign_func(111, F.STR_MATCH, [1, 101]).assign_func(112, F.STR_LENGTH, [1])
         .filter(110).filter(111).group_by([Aggregate(120, AGG.COUNT, []), Aggregate(121, AGG.SUM, [112])], keys=[3, 112]))
    got, exp, (on_codes, decoded) = both(RES, p, batch_of(rng, 50_000, nulls=0.05))
    compare_batches(got, exp, key_names=["engine", "112"])
    assert decoded == 0 and on_codes == 1


def test_uses_that_need_the_strings_decode():
    rng = np.random.default_rng(9700)
    b = batch_of(rng, 12_000, nulls=0.1)
    # the column as a value: projection, MIN / MAX / SOME, a compare of two columns
    p1 = Program().assign_const(100, "Int64", 0).assign_func(110, F.CMP_GREATER, [2, 100]).filter(110).project([1, 2, 3])
    p2 = Program().group_by([Aggregate(120, AGG.MIN, [1]), Aggregate(121, AGG.MAX, [1]), Aggregate(122, AGG.SOME, [3])], keys=[2])
    p3 = Program().assign_func(110, F.CMP_LESS, [1, 3]).project([110])
    for p, keys in ((p1, None), (p2, ["v"]), (p3, None)):
        got, exp, (on_codes, decoded) = both(RES, p, b)
        compare_batches(got, exp, key_names=keys)
        assert decoded >= 1


def test_mixed_plain_and_dictionary_keys():
    rng = np.random.default_rng(9800)
    n = 25_000
    b = batch_of(rng, n, nulls=0.1)
    b["plain"] = ["p%d" % k for k in rng.integers(0, 6, n)]
    res = RES + [(4, "plain")]
    p = Program().group_by([Aggregate(120, AGG.COUNT, []), Aggregate(121, AGG.SUM, [2])], keys=[1, 4, 3, 2])
    got, exp, _ = both(res, p, b)
    compare_batches(got, exp, key_names=["phrase", "plain", "engine", "v"])


def test_filter_built_on_codes():
    from oracle import ssa_oracle
    from ydb_amd.program import ProgramContainer
    rng = np.random.default_rng(9900)
    b = batch_of(rng, 33_333, nulls=0.1)
    p = (Program().assign_const(100, "Bytes", b"a").assign_yql(110, YQL.StringContains, [1, 100]).filter(110)
         .assign_const(101, "Int64", 3).assign_func(111, F.CMP_LESS, [2, 101]).filter(111).project([2]))
    pc = ProgramContainer()
    ok, err = pc.Init(RES, p)
    assert ok, err
    from ydb_amd.column_filter import ColumnFilter
    mask, selected = pc.BuildFilter(b)
    exp = np.asarray(ssa_oracle.Oracle(RES, p).build_filter(b)).astype(bool)
    assert selected == int(np.count_nonzero(exp))
    assert np.array_equal(np.asarray(mask).astype(bool), exp)
    s1 = stats()
    f = ColumnFilter.FromProgram(pc, pc.Import(b), 0)
    assert f.GetFilteredCount() == selected and np.array_equal(f.BuildSimpleFilter(), mask)
    s2 = stats()
    assert s2[0] - s1[0] == 1 and s2[1] - s1[1] == 0


def test_sliced_dictionary_column():
    rng = np.random.default_rng(9950)
    total, off = 9000, 41
    codes = rng.integers(0, len(WORDS), total).astype(np.int16)
    valid = rng.random(total) > 0.2
    v = rng.integers(0, 5, total).astype(np.int64)
    p = Program().group_by([Aggregate(120, AGG.COUNT, []), Aggregate(121, AGG.SUM, [2])], keys=[1])
    b = {"phrase": Col(codes, valid=valid, offset=off, dictionary=WORDS), "v": Col(v, offset=off)}
    got, exp, _ = both(RES[:2], p, b)
    compare_batches(got, exp, key_names=["phrase"])


def test_device_resident_dictionary_column():
    import torch
    from ydb_amd import _lib
    from ydb_amd.program import ProgramContainer, device_batch
    from oracle import ssa_oracle
    rng = np.random.default_rng(9990)
    n = 200_000
    words = [b"N", b"R", b"A"]
    codes = rng.integers(0, 3, n).astype(np.uint8)
    qty = rng.integers(1, 51, n).astype(np.float64)
    dev = torch.device("cuda:0")
    t_codes = torch.from_numpy(codes).to(dev)
    t_qty = torch.from_numpy(qty).to(dev)
    chars = np.frombuffer(b"".join(words) + b"\0" * 16, np.uint8).copy()
    t_chars = torch.from_numpy(chars).to(dev)
    t_offs = torch.tensor([0, 1, 2, 3], dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    cols = [("flag", 14, t_codes.data_ptr(), 0, t_chars.data_ptr(), 3, 0, 6, t_offs.data_ptr(), 3),
            ("qty", 11, t_qty.data_ptr(), 0, 0, 0, 0)]
    batch = device_batch(cols, n, keep=(t_codes, t_qty, t_chars, t_offs))
    p = Program().group_by([Aggregate(120, AGG.SUM, [2]), Aggregate(121, AGG.COUNT, [])], keys=[1])
    res = [(1, "flag"), (2, "qty")]
    pc = ProgramContainer()
    ok, err = pc.Init(res, p)
    assert ok, err
    s0 = stats()
    got = pc.Export(pc.ApplyDevice(batch))
    s1 = stats()
    exp = ssa_oracle.Oracle(res, p).apply({"flag": Col(codes, dictionary=words, type="string"), "qty": qty})
    compare_batches(got, exp, key_names=["flag"], float_sum_names=["120"])
    assert s1[0] - s0[0] == 1 and s1[1] - s0[1] == 0


def test_q1_over_dictionary_flags_equals_q1_over_utf8_flags():
    """TPC-H Q1 on a device-generated lineitem: the flags as dictionary columns (1-byte codes + 'A' 'N' 'R' / 'F' 'O')
    against the same flags as utf8 columns, through the kernel built for the program (64 M rows)."""
    from ydb_amd import tpch
    from ydb_amd.program import ProgramContainer
    rows = 64_000_000
    li = tpch.DeviceLineitem(rows)
    prog = tpch.q1_program(False)
    res = {}
    for name, cols in (("utf8", tpch.Q1_COLUMNS_UTF8), ("dict", tpch.Q1_COLUMNS_DICT)):
        pc = ProgramContainer()
        ok, err = pc.Init(tpch.LINEITEM_RESOLVER, prog)
        assert ok, err
        s0 = stats()
        for _ in range(3):  # generic kernel, built kernel, built kernel with its launch settled
            res[name] = pc.Export(pc.ApplyDevice(li.batch(cols)))
        s1 = stats()
        if name == "dict":
            assert s1[0] - s0[0] == 3 and s1[1] == s0[1]
    compare_batches(res["dict"], res["utf8"], key_names=["l_returnflag", "l_linestatus"],
                    float_sum_names=["130", "131", "132", "133", "136"], rtol=1e-12)
    assert int(np.sum(res["dict"]["138"].values)) == int(np.sum(res["utf8"]["138"].values))

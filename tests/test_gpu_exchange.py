"""Result batches between GPUs: pack / unpack / concat / slice on one GPU against the host's view of the same batches, and
the whole N > 1 path rehearsed with two ranks on one GPU (child processes over gloo)."""
import os
import pickle
import socket
import subprocess
import sys

import numpy as np
import pytest

from cases import compare_batches
from ydb_amd.arrow_c import Col
from ydb_amd.ssa_pb import AGG, Aggregate, Program

pytestmark = pytest.mark.gpu

RES = [(1, "k"), (2, "s"), (3, "v"), (4, "d")]


def part(rng, n, nulls):
    words = ["zz", "a", "bb", "a longer string than the rest of them", "ccc"]
    v = rng.integers(-100, 100, n).astype(np.int64)
    return {"k": rng.integers(0, 50, n).astype(np.int32),
            "s": Col([words[i] for i in rng.integers(0, len(words), n)], type="string", valid=(rng.random(n) > 0.2) if nulls else None),
            "v": Col(v, valid=(rng.random(n) > 0.3) if nulls else None),
            "d": rng.random(n)}


def host_concat(parts):
    out = {}
    for name in parts[0]:
        vals, valid = [], []
        for p in parts:
            c = p[name]
            values = getattr(c, "values", c)
            ok = getattr(c, "valid", None)
            n = len(values)
            valid.extend([True] * n if ok is None else list(ok))
            vals.extend(list(values))
        out[name] = (vals, valid)
    return out


def check_equal(exported, expected):
    for name, (vals, valid) in expected.items():
        got = exported[name]
        assert list(np.asarray(got.valid, dtype=bool)) == [bool(x) for x in valid], name
        gv = got.values
        for i, ok in enumerate(valid):
            if not ok:
                continue
            g = gv[i]
            e = vals[i]
            if isinstance(e, str):
                e = e.encode()
            assert g == e, (name, i, g, e)


def test_pack_unpack_concat_slice_round_trip():
    import torch
    from ydb_amd.program import ProgramContainer
    rng = np.random.default_rng(77)
    pc = ProgramContainer()
    ok, err = pc.Init(RES, Program().project([1, 2, 3, 4]))
    assert ok, err
    host_parts = [part(rng, 1000, True), part(rng, 0, False), part(rng, 1, False), part(rng, 4097, True), part(rng, 300, False)]
    dev_parts = [pc.Import(p) for p in host_parts]
    # through the flat layout and back
    unpacked, keep = [], []
    for b in dev_parts:
        header, data_bytes = pc.PackLayout(b)
        buf = torch.empty(data_bytes + 512, dtype=torch.uint8, device="cuda:0")
        pc.PackData(b, buf.data_ptr(), data_bytes)
        host = buf.cpu()  # as a collective over the host would: the bytes land somewhere else
        again = host.to("cuda:0")
        keep.append(again)
        unpacked.append(pc.Unpack(header, again.data_ptr(), data_bytes, keep=again))
    for u, h in zip(unpacked, host_parts):
        check_equal(pc.Export(u), host_concat([h]))
    whole = pc.Concat(unpacked)
    assert whole.rows == sum(len(p["k"]) for p in host_parts)
    expected = host_concat(host_parts)
    check_equal(pc.Export(whole), expected)
    # views of the whole
    for row0, rows in ((0, 0), (0, 1000), (999, 3), (1001, 4097), (5000, 398)):
        sl = pc.Export(pc.Slice(whole, row0, rows))
        check_equal(sl, {k: (v[0][row0:row0 + rows], v[1][row0:row0 + rows]) for k, v in expected.items()})
    # a program over a concatenated batch of views
    from oracle import ssa_oracle
    p = Program().group_by([Aggregate(120, AGG.COUNT, [3]), Aggregate(121, AGG.SUM, [3]), Aggregate(122, AGG.SUM, [4])], keys=[1, 2])
    pc2 = ProgramContainer()
    ok, err = pc2.Init(RES, p)
    assert ok, err
    got = pc2.Export(pc2.ApplyDevice(pc.Slice(whole, 500, 4000)))
    # (a null row of a string column keeps its bytes, and the grouping looks at the bytes: the oracle gets them too)
    flat = {k: Col(list(v[0][500:4500]), type="string", valid=np.asarray(v[1][500:4500], dtype=bool)) if k == "s"
            else Col(np.asarray(v[0][500:4500]), valid=np.asarray(v[1][500:4500], dtype=bool)) for k, v in expected.items()}
    exp = ssa_oracle.Oracle(RES, p).apply(flat)
    compare_batches(got, exp, key_names=["k", "s"], float_sum_names=["122"])


def test_concat_of_boolean_columns_from_views_at_any_bit():
    """Boolean values are bits: the parts of a concatenation start at any bit of their bitmaps (views of other batches,
    results of compare steps), with and without nulls."""
    from ydb_amd.program import ProgramContainer
    rng = np.random.default_rng(79)
    res = [(1, "b"), (2, "k"), (3, "c")]
    pc = ProgramContainer()
    ok, err = pc.Init(res, Program().project([1, 2, 3]))
    assert ok, err

    def bools(n, nulls):
        return {"b": Col(rng.random(n) > 0.5, valid=(rng.random(n) > 0.3) if nulls else None), "k": rng.integers(0, 9, n).astype(np.int16),
                "c": rng.random(n) > 0.1}
    host = [bools(1000, True), bools(0, False), bools(1, False), bools(4097, True), bools(300, False)]
    cuts = [(3, 900), (0, 0), (0, 1), (13, 4001), (7, 293)]
    views = [pc.Slice(pc.Import(h), at, n) for h, (at, n) in zip(host, cuts)]
    whole = pc.Concat(views)
    assert whole.rows == sum(n for _, n in cuts)

    def cut(h, at, n):
        return {k: (Col(v.values[at:at + n], valid=None if v.valid is None else v.valid[at:at + n]) if isinstance(v, Col) else v[at:at + n])
                for k, v in h.items()}
    expected = host_concat([cut(h, at, n) for h, (at, n) in zip(host, cuts)])
    check_equal(pc.Export(whole), expected)
    # through the flat layout as well, and once more concatenated with itself
    import torch
    header, data_bytes = pc.PackLayout(whole)
    buf = torch.empty(data_bytes + 64, dtype=torch.uint8, device="cuda:0")
    pc.PackData(whole, buf.data_ptr(), data_bytes)
    again = pc.Unpack(header, buf.data_ptr(), data_bytes, keep=buf)
    twice = pc.Export(pc.Concat([again, pc.Slice(whole, 5, 100)]))
    check_equal(twice, {k: (v[0] + v[0][5:105], v[1] + v[1][5:105]) for k, v in expected.items()})


def test_damaged_headers_are_refused():
    import torch
    from ydb_amd.program import ProgramContainer, SsaError
    rng = np.random.default_rng(78)
    pc = ProgramContainer()
    ok, err = pc.Init(RES, Program().project([1, 2, 3, 4]))
    assert ok, err
    b = pc.Import(part(rng, 100, True))
    header, data_bytes = pc.PackLayout(b)
    buf = torch.zeros(data_bytes, dtype=torch.uint8, device="cuda:0")
    pc.PackData(b, buf.data_ptr(), data_bytes)
    with pytest.raises(SsaError):
        pc.Unpack(header[:20], buf.data_ptr(), data_bytes, keep=buf)
    with pytest.raises(SsaError):
        pc.Unpack(b"XXXX" + header[4:], buf.data_ptr(), data_bytes, keep=buf)
    with pytest.raises(SsaError):
        pc.Unpack(header, buf.data_ptr(), data_bytes // 2, keep=buf)
    with pytest.raises(SsaError):
        pc.PackData(b, buf.data_ptr(), 16)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_ranks_on_one_gpu_end_with_the_whole_tables_result(tmp_path):
    from ydb_amd import tpch
    from ydb_amd.program import ProgramContainer
    here = os.path.dirname(os.path.abspath(__file__))
    rows = 1_500_000
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(here, "helpers", "rehearse_exchange.py"), str(tmp_path), str(rows)]
    done = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
    assert done.returncode == 0, done.stdout[-3000:] + done.stderr[-3000:]
    ranks = []
    for r in range(2):
        with open(tmp_path / f"rank{r}.pkl", "rb") as f:
            ranks.append(pickle.load(f))
    sys.path.insert(0, os.path.join(here, "helpers"))
    import rehearse_exchange
    li = tpch.DeviceLineitem(2 * rows)
    for name, (prog, cols) in rehearse_exchange.programs().items():
        pc = ProgramContainer()
        ok, err = pc.Init(tpch.LINEITEM_RESOLVER, prog)
        assert ok, err
        whole = pc.Export(pc.ApplyDevice(li.batch(cols)))
        n_groups = len(next(iter(whole.values())).valid)
        key_names = [k for k in whole if not k.isdigit()]

        def rows_of(res):
            """{key tuple: tuple of the other columns}"""
            names = list(res)
            n = len(res[names[0]][2])
            out = {}
            for i in range(n):
                key = tuple((res[k][1][i] if res[k][2][i] else None) for k in key_names)
                out[key] = tuple((res[k][1][i] if res[k][2][i] else None) for k in names if k not in key_names)
            return names, out

        want_names, want = rows_of({k: (c.type, list(c.values) if isinstance(c.values, list) else np.asarray(c.values), np.asarray(c.valid)) for k, c in whole.items()})
        assert len(want) == n_groups
        if name != "many_groups":
            # the mixed stream (Q1 and Q6 alternating over the resident shard): every step of it ends, on every rank, with
            # the whole table's result
            for step in range(3):
                for r in ranks:
                    _, t = rows_of(r[("mixed", step, name)])
                    assert set(t) == set(want), (name, step)
                    for key, vals in want.items():
                        for g, e in zip(t[key], vals):
                            if isinstance(e, (float, np.floating)):
                                assert abs(g - e) <= 1e-11 * max(1.0, abs(e)), (name, step, key, g, e)
                            else:
                                assert g == e, (name, step, key, g, e)
        for exchange in (("allgather", "alltoall", "auto") if name != "q6" else ("allgather",)):
            parts = [rows_of(r[(name, exchange)]) for r in ranks]
            resolved = exchange
            if exchange == "auto":
                resolved = "alltoall" if name == "many_groups" else "allgather"
            if resolved == "allgather":  # every rank has the whole result
                tables = [p[1] for p in parts]
            else:  # the ranks' parts are disjoint and together the whole result
                assert not (set(parts[0][1]) & set(parts[1][1])), (name, exchange)
                if name == "many_groups":
                    assert min(len(parts[0][1]), len(parts[1][1])) > n_groups // 4
                tables = [{**parts[0][1], **parts[1][1]}]
            for t in tables:
                assert set(t) == set(want), (name, exchange)
                for key, vals in want.items():
                    got = t[key]
                    for g, e, col in zip(got, vals, [c for c in want_names if c not in key_names]):
                        if isinstance(e, float) or isinstance(e, np.floating):
                            assert abs(g - e) <= 1e-11 * max(1.0, abs(e)), (name, exchange, key, col, g, e)
                        else:
                            assert g == e, (name, exchange, key, col, g, e)


def test_two_ranks_run_the_mixed_stream_of_the_benchmark():
    """bench.py --workload mixed with two ranks on one GPU (the exchange over gloo): one JSON line from rank 0 that counts
    both ranks' rows of both programs."""
    import json
    here = os.path.dirname(os.path.abspath(__file__))
    rows = 2_000_000
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(here, "..", "bench.py"), "--gpus", "2", "--workload", "mixed", "--rows", str(rows),
           "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--rehearse-on-one-gpu"]
    done = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
    assert done.returncode == 0, done.stdout[-3000:] + done.stderr[-3000:]
    lines = [ln for ln in done.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, done.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak"
    assert "mixed" in line["config"]["workload"] or "Q1" in line["config"]["workload"]
    # every step scans the shard once per program on either rank
    assert abs(line["value"] * line["ms_per_step"] / 1e3 - 2 * 2 * rows) <= 1e-6 * 4 * rows


def test_one_rank_over_rccl_runs_the_device_exchange():
    """The collectives of the exchange on device buffers over the `nccl` backend (RCCL), which a box with one GPU can run
    with one rank only: all three exchange settings end with the plain scan's result."""
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    done = subprocess.run([sys.executable, os.path.join(here, "helpers", "rccl_one_rank.py"), "600000"], env=env, capture_output=True,
                          text=True, timeout=420)
    assert done.returncode == 0, done.stdout[-3000:] + done.stderr[-3000:]
    assert "exchanges equal the plain scan" in done.stdout

"""Host threads over one device (the reference runs one batch per Conveyor worker thread, each with the thread-local
context of ydb/core/formats/arrow/custom_registry.cpp:101-105), kernels built in the background, the code-object cache
on disk, and the device pool after a failed allocation."""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from cases import compare_batches

pytestmark = pytest.mark.gpu


def test_eight_threads_with_their_own_handles():
    from oracle import ssa_oracle
    from ydb_amd import tpch
    from ydb_amd.program import ProgramContainer
    from ydb_amd.ssa_pb import AGG, F, Aggregate, Program
    n = 300_000
    host = tpch.lineitem_host(0, n)
    programs = []
    for t in range(8):   # eight different programs: different constants, keyed / key-less / projection
        if t % 3 == 0:
            p, cols, keys, sums = tpch.q1_program(False), tpch.Q1_COLUMNS_UTF8, ["l_returnflag", "l_linestatus"], ["130", "131", "132", "133", "136"]
        elif t % 3 == 1:
            p = (Program().assign_const(100, "Int32", 9000 + 100 * t).assign_func(110, F.CMP_GREATER_EQUAL, [tpch.L_SHIPDATE, 100]).filter(110)
                 .group_by([Aggregate(130, AGG.SUM, [tpch.L_QUANTITY]), Aggregate(131, AGG.COUNT, []), Aggregate(132, AGG.MAX, [tpch.L_DISCOUNT])]))
            cols, keys, sums = ["l_shipdate", "l_quantity", "l_discount"], [], ["130"]
        else:
            p = (Program().assign_const(100, "Double", 10.0 + t).assign_func(110, F.CMP_LESS, [tpch.L_QUANTITY, 100]).filter(110)
                 .assign_func(120, F.MATH_MULTIPLY, [tpch.L_EXTENDEDPRICE, tpch.L_DISCOUNT]).project([120, tpch.L_SHIPDATE]))
            cols, keys, sums = ["l_quantity", "l_extendedprice", "l_discount", "l_shipdate"], [], []
        batch = {c: host[c] for c in cols}
        exp = ssa_oracle.Oracle(tpch.LINEITEM_RESOLVER, p).apply(batch)
        programs.append((p, batch, keys, sums, exp))
    errors = []
    start = threading.Barrier(8)

    def work(i):
        try:
            p, batch, keys, sums, exp = programs[i]
            pc = ProgramContainer()
            ok, err = pc.Init(tpch.LINEITEM_RESOLVER, p)
            assert ok, err
            dev = pc.Import(batch)
            start.wait()
            for _ in range(12):
                got = pc.Export(pc.ApplyDevice(dev))
                compare_batches(got, exp, key_names=keys, float_sum_names=sums, rtol=1e-11)
                got = pc.ApplyProgram(batch)   # the host seam: import, apply, export on this thread
                compare_batches(got, exp, key_names=keys, float_sum_names=sums, rtol=1e-11)
        except BaseException as e:  # noqa: BLE001 - reported by the main thread
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_threads_sharing_one_handle_are_serialised():
    from oracle import ssa_oracle
    from ydb_amd import tpch
    from ydb_amd.program import ProgramContainer
    n = 200_000
    host = tpch.lineitem_host(0, n)
    batch = {c: host[c] for c in tpch.Q6_COLUMNS}
    exp = ssa_oracle.Oracle(tpch.LINEITEM_RESOLVER, tpch.q6_program()).apply(batch)
    pc = ProgramContainer()
    assert pc.Init(tpch.LINEITEM_RESOLVER, tpch.q6_program())[0]
    errors = []

    def work():
        try:
            for _ in range(10):
                compare_batches(pc.ApplyProgram(batch), exp, float_sum_names=["130"], rtol=1e-11)
        except BaseException as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=work) for _ in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


_CHILD = r"""
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from ydb_amd import tpch
from ydb_amd.program import ProgramContainer
mode = sys.argv[1]
n = 4_000_000
li = tpch.DeviceLineitem(n, utf8_flags=False, code_flags=False)
pc = ProgramContainer()
assert pc.Init(tpch.LINEITEM_RESOLVER, tpch.q6_program())[0]
b = li.batch(tpch.Q6_COLUMNS)
first = pc.Export(pc.ApplyDevice(b))["130"].values[0]
waits = []
deadline = time.time() + 60
applies = 0
while time.time() < deadline:
    t0 = time.perf_counter()
    v = pc.Export(pc.ApplyDevice(b))["130"].values[0]
    waits.append(time.perf_counter() - t0)
    applies += 1
    assert abs(v - first) <= 1e-12 * abs(first), (v, first)
    built, ms, launches = pc.SpecializationStats()
    if launches > 0 and applies > 12:
        break
built, ms, launches = pc.SpecializationStats()
print("RESULT", mode, built, round(ms), launches, applies, round(max(waits) * 1e3, 2))
"""


def _child(mode, env_extra):
    env = dict(os.environ)
    env.pop("SSA_JIT", None)
    env.update(env_extra)
    out = subprocess.run([sys.executable, "-c", _CHILD, mode], env=env, capture_output=True, text=True, timeout=240,
                         cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT")][-1].split()
    return {"built": int(line[2]), "build_ms": float(line[3]), "launches": int(line[4]), "applies": int(line[5]), "max_wait_ms": float(line[6])}


def test_batches_of_a_shard_never_wait_for_a_build(tmp_path):
    # 4 M-row batches reach the build threshold (2^25 rows) at the ninth scan: a worker thread builds, the calling
    # thread keeps scanning with the generic kernel and no apply takes anywhere near the build time
    r = _child("async", {})
    assert r["built"] >= 1 and r["launches"] > 0, r
    assert r["max_wait_ms"] < 0.5 * r["build_ms"], r
    # SSA_JIT_ASYNC=0: the ninth scan waits for its kernel
    r = _child("sync", {"SSA_JIT_ASYNC": "0"})
    assert r["built"] >= 1 and r["max_wait_ms"] > 0.5 * r["build_ms"], r
    # a cache directory: the second process loads the code object instead of compiling
    d = str(tmp_path)
    a = _child("disk1", {"SSA_JIT_CACHE_DIR": d, "SSA_JIT": "1"})
    assert a["build_ms"] > 50 and any(f.endswith(".co") for f in os.listdir(d)), (a, os.listdir(d))
    b = _child("disk2", {"SSA_JIT_CACHE_DIR": d, "SSA_JIT": "1"})
    assert b["built"] >= 1 and b["build_ms"] == 0 and b["launches"] > 0, b
    # ... but only what a library of the same build left there: the key of a code object holds a digest of the kernel
    # sources, the compiler's options and the runtime's version (SSA_JIT_KEY_SALT stands in for another build)
    c = _child("disk3", {"SSA_JIT_CACHE_DIR": d, "SSA_JIT": "1", "SSA_JIT_KEY_SALT": "another build"})
    assert c["built"] >= 1 and c["build_ms"] > 50 and c["launches"] > 0, c


def test_a_failed_allocation_does_not_poison_the_next_launch():
    # the n-th hipMalloc of a fresh process fails once (as on a full device): the pool trims and tries again, and the
    # scan that follows must not see the failed call's error
    code = r"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from ydb_amd import tpch
from ydb_amd.program import ProgramContainer
host = tpch.lineitem_host(0, 50_000)
pc = ProgramContainer()
assert pc.Init(tpch.LINEITEM_RESOLVER, tpch.q6_program())[0]
a = pc.ApplyProgram({c: host[c] for c in tpch.Q6_COLUMNS})["130"].values[0]
b = pc.ApplyProgram({c: host[c] for c in tpch.Q6_COLUMNS})["130"].values[0]
assert a == b
print("OK", a)
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    good = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, cwd=root)
    assert good.returncode == 0, good.stderr[-2000:]
    for nth in (2, 3, 5, 8):
        env = dict(os.environ, SSA_TEST_FAIL_ALLOC=str(nth))
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120, cwd=root)
        assert out.returncode == 0, (nth, out.stderr[-2000:])
        assert out.stdout.strip().splitlines()[-1] == good.stdout.strip().splitlines()[-1]

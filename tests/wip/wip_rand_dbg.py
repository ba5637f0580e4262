import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np
import importlib.util
spec = importlib.util.spec_from_file_location("rp", "tests/test_gpu_random_programs.py"); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
from oracle import ssa_oracle
from ydb_amd.program import ProgramContainer, SsaError
def show(p):
    for c in p.commands:
        t = type(c).__name__
        if t == "Assign":
            f = c.function
            print("    A", c.column, ("f%d yql=%s args=%s" % (f.id, f.yql_op, list(f.args))) if f else ("const " + str(c.constant)))
        else:
            print("    ", c)
cases = [("g", 65), ("g", 66), ("p", 69), ("p", 83)]
for kind, seed in cases:
    if kind == "g":
        rng = np.random.default_rng(5000 + seed); n = int(rng.choice([1, 64, 1025, 4097, 50_021])); b = m.make_batch(rng, n)
        p, keys, fs = m.random_group_by(rng, int(rng.integers(0, 6)))
    else:
        rng = np.random.default_rng(9000 + seed); n = int(rng.choice([1, 64, 1025, 4097, 30_011])); b = m.make_batch(rng, n)
        p = m.random_filter_projection(rng, int(rng.integers(1, 8)))
    print(kind, seed, "n", n)
    show(p)
    try:
        exp = ssa_oracle.Oracle(m.RES, p).apply(b); print("  oracle ok rows", len(next(iter(exp.values())).valid))
    except ssa_oracle.OracleError as e:
        print("  oracle error", e)
    os.environ["SSA_JIT"] = "0"; os.environ["SSA_DEBUG_PLAN"] = "1"
    pc = ProgramContainer(); ok, err = pc.Init(m.RES, p)
    try:
        got = pc.ApplyProgram(b); print("  gpu ok rows", len(next(iter(got.values())).valid))
    except SsaError as e:
        print("  gpu error", e)

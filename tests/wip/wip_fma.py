import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
os.environ["SSA_JIT"] = sys.argv[1] if len(sys.argv) > 1 else "1"
import numpy as np
from ydb_amd.ssa_pb import Program, F, Column
from ydb_amd.program import ProgramContainer
from oracle import ssa_oracle
rng = np.random.default_rng(1)
n = 200_000
b = {"a": rng.normal(0, 1, n), "b": rng.normal(0, 1, n), "c": rng.normal(0, 1, n)}
res = [(1, "a"), (2, "b"), (3, "c")]
for name, p in {
    "arrow a*b+c": Program().assign_func(100, F.MATH_MULTIPLY, [1, 2]).assign_func(101, F.MATH_ADD, [100, 3]).project([101]),
    "yql a*b-c": Program(version=5).assign_yql(100, 5, [1, 2]).assign_yql(101, 4, [100, 3]).project([101]),
}.items():
    pc = ProgramContainer(); ok, err = pc.Init(res, p); assert ok, err
    got = pc.ApplyProgram(b)["101"].to_pylist()
    exp = ssa_oracle.Oracle(res, p).apply(b)["101"].to_pylist()
    bad = sum(1 for x, y in zip(got, exp) if x != y)
    print(name, "SSA_JIT", os.environ["SSA_JIT"], "mismatches", bad, "of", n)

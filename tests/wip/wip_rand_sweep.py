"""One-off wider sweep of the random-program generators (other seeds than the committed tests), generic kernels only."""
import os, sys
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
os.environ["SSA_JIT"] = "0"
import numpy as np
import importlib.util
spec = importlib.util.spec_from_file_location("rp", "tests/test_gpu_random_programs.py"); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
from cases import compare_batches
from oracle import ssa_oracle
from ydb_amd.program import ProgramContainer, SsaError
bad = 0
def one(tag, seed, p, b, keys=(), fs=()):
    global bad
    pc = ProgramContainer(); ok, err = pc.Init(m.RES, p)
    got = gerr = exp = oerr = None
    try: got = pc.ApplyProgram(b)
    except SsaError as e: gerr = str(e)
    try: exp = ssa_oracle.Oracle(m.RES, p).apply(b)
    except ssa_oracle.OracleError as e: oerr = str(e)
    if oerr is not None:
        if gerr is None: bad += 1; print("MISS", tag, seed, "oracle:", oerr)
        return
    if gerr is not None:
        if "unsupported" not in gerr.lower(): bad += 1; print("GPUERR", tag, seed, gerr)
        return
    try: compare_batches(got, exp, key_names=list(keys), float_sum_names=list(fs))
    except AssertionError as e: bad += 1; print("DIFF", tag, seed, str(e)[:200])
lo, hi = int(sys.argv[1]), int(sys.argv[2])
for seed in range(lo, hi):
    rng = np.random.default_rng(5000 + seed); n = int(rng.choice([1, 64, 1025, 4097, 50_021])); b = m.make_batch(rng, n)
    p, keys, fs = m.random_group_by(rng, int(rng.integers(0, 6))); one("g", seed, p, b, keys, fs)
    rng = np.random.default_rng(9000 + seed); n = int(rng.choice([1, 64, 1025, 4097, 30_011])); b = m.make_batch(rng, n)
    p = m.random_filter_projection(rng, int(rng.integers(1, 8))); one("p", seed, p, b)
    rng = np.random.default_rng(1000 + seed); n = int(rng.choice([1, 63, 1025, 20_011])); b = m.make_batch(rng, n)
    p = m.random_program(rng, int(rng.integers(2, 9)))
    if p is not None: one("e", seed, p, b)
    if seed % 50 == 0: print("at", seed, flush=True)
print("done", lo, hi, "bad", bad)

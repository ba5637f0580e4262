import sys, os
sys.path.insert(0, "tests")
import numpy as np
from test_gpu_parity import *
n = 1
rng = np.random.default_rng(2000 + n)
b = lineitem_like(rng, n, 0.0)
got, exp = both(LI, q1_like([5, 6]), b)
for k in exp:
    print(os.environ.get("SSA_JIT"), n, k, got[k].to_pylist()[:4], exp[k].to_pylist()[:4])

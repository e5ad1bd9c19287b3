"""Phase clocks of the band + border LDL^T (experiment build -DIPM_TIMING, perf exploration only):
make -C lpopc_amd/csrc librpm_exp_ipmt.so EXPFLAGS=-DIPM_TIMING;  RPM_HIP_LIB=.../librpm_exp_ipmt.so python tools/ipm_factor_timing.py [instances]"""
import os
import sys
import time

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
os.environ.setdefault("RPM_HIP_LIB", os.path.join(root, "lpopc_amd", "csrc", "librpm_exp_ipmt.so"))
import numpy as np
import torch  # noqa: F401

from lpopc_amd import problems
from lpopc_amd.engine import BatchedIPM, NLPEngine
from test_ipm import _exact, _random_kkt

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
eng = NLPEngine(problems.quadrotor(8, 8), _exact(), n_instances=B, device=0)
ipm = BatchedIPM(eng)
store, dense, sign = _random_kkt(ipm, eng.n, 1, 7)
store = np.tile(store, (B, 1))
rhs = np.random.RandomState(3).uniform(-1, 1, size=(B, sign.size))
for rep in range(2):
    t0 = time.perf_counter()
    sol, npos, nneg = ipm.debug_solve(store, rhs)
    print("debug_solve wall %.3f s" % (time.perf_counter() - t0), flush=True)
ref = np.linalg.solve(dense[0], rhs[0])
print("rel err", np.max(np.abs(sol[0] - ref)) / np.max(np.abs(ref)))

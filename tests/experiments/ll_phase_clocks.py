"""Phase clocks of the left-looking kkt_factor_kernel on an interval block of the metric problem's size (16 LGR points), experiment build:
make -C lpopc_amd/csrc librpm_exp_ipmt.so EXPFLAGS="-DIPM_TIMING -DIPM_TIMING_SUB=0"
RPM_HIP_LIB=lpopc_amd/csrc/librpm_exp_ipmt.so python tests/experiments/ll_phase_clocks.py"""
import os
import sys
import time

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import torch  # noqa: F401

from lpopc_amd import problems
from lpopc_amd.engine import BatchedIPM, NLPEngine
from test_ipm import _exact, _random_kkt_dense

eng = NLPEngine(problems.launch(2, 16), _exact(), n_instances=1, device=0)
eng.set_option("ipm_nested", 1)
ipm = BatchedIPM(eng)
print(ipm.info(), ipm.subproblems()[:3], flush=True)
t0 = time.time()
dense, sign, filled = _random_kkt_dense(ipm, eng.n, 1, 23)
print("matrix built in %.1f s" % (time.time() - t0), flush=True)
rhs = np.random.RandomState(9).uniform(-1, 1, size=(1, sign.size))
for rep in range(2):
    sol, npos, nneg = ipm.debug_solve_dense(dense, rhs)
ref = np.linalg.solve(dense[0], rhs[0])
print("rel err", np.max(np.abs(sol[0] - ref)) / np.max(np.abs(ref)))

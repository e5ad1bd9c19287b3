"""Phase clocks of kkt_factor_dense_kernel on the sweep's interval blocks (experiment build, perf exploration only):
make -C lpopc_amd/csrc librpm_exp_ipmt.so EXPFLAGS="-DIPM_TIMING -DIPM_TIMING_SUB=100000" [more -D...]
RPM_HIP_LIB=lpopc_amd/csrc/librpm_exp_ipmt.so python tests/experiments/dense_phase_clocks.py [instances] [launch]
(launch: the metric problem's interval blocks, 19 block rows, on a 4 x 2 x 16 mesh)"""
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

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
eng = NLPEngine(problems.launch(2, 16) if "launch" in sys.argv[2:] else problems.quadrotor(8, 8), _exact(), n_instances=B, device=0)
eng.set_option("ipm_nested", 1)
ipm = BatchedIPM(eng)
dense, sign, filled = _random_kkt_dense(ipm, eng.n, 1, 23)
dense = np.tile(dense, (B, 1, 1))
rhs = np.random.RandomState(9).uniform(-1, 1, size=(B, sign.size))
for rep in range(2):
    t0 = time.perf_counter()
    sol, npos, nneg = ipm.debug_solve_dense(dense, rhs)
    print("debug_solve_dense wall %.3f s" % (time.perf_counter() - t0), flush=True)
ref = np.linalg.solve(dense[0], rhs[0])
print("rel err", np.max(np.abs(sol[0] - ref)) / np.max(np.abs(ref)))

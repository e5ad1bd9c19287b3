"""Row f-2 on the METRIC problem (Delta-III, 4 phases x K intervals x Nk LGR points): wall time of one device
interior-point iteration (callbacks + KKT assembly + band/border LDL^T + substitution + line search), whatever the solve's
final status — the solver does not converge on this problem from lpopc's default guess (DESIGN.md f-2 lists why), but an
iteration is an iteration.  Prints one JSON object.  Run on the GPU box: python tools/ipm_config3_iter.py [K] [Nk] [iters] [nested 0|1]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from lpopc_amd import problems
from lpopc_amd.engine import BatchedIPM, NLPEngine
from lpopc_amd.problem import Options

K = int(sys.argv[1]) if len(sys.argv) > 1 else 64
Nk = int(sys.argv[2]) if len(sys.argv) > 2 else 16
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 6
nested = int(sys.argv[4]) if len(sys.argv) > 4 else 1
o = Options()
o.SetStringValue("hessian-approximation", "exact")
prob = problems.launch(K, Nk)
eng = NLPEngine(prob, o, device=0)
eng.set_option("ipm_nested", nested)
t0 = time.perf_counter()
ipm = BatchedIPM(eng, max_iter=iters, restoration=0)
t_create = time.perf_counter() - t0
x0 = eng.get_starting_point()[None, :]
t0 = time.perf_counter()
r = ipm.solve(x0)
dt = time.perf_counter() - t0
st, info, kt = ipm.stats(), ipm.info(), ipm.kernel_times()
print(json.dumps({"problem": "Delta-III 4 x %d x %d" % (K, Nk), "factorisation": "nested dissection over the mesh intervals" if nested else "band + border", "n": eng.n, "m": eng.m, "nnz_h": eng.nnz_h, "kkt": info,
                  "kkt_storage_mb": info["storage_doubles"] * 8 / 1e6, "create_s": t_create, "solve_s": dt,
                  "iterations": st["iterations"], "factorizations": st["factorizations"], "trial_points": st["trial_points"],
                  "ms_per_ipm_iteration": 1e3 * dt / max(1, st["iterations"]),
                  "factor_ms_per_launch": kt["factor_ms"] / max(1, st["factorizations"]),
                  "substitution_ms_total": kt["substitution_ms"], "status": int(r["status"][0]), "kkt_error": float(r["kkt_error"][0])}))

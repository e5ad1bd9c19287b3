"""Row f-2 on the metric problem: solve Delta-III (4 phases x K intervals x Nk LGR points) on the device from lpopc's default
guess.  Prints one JSON object (status 0 converged, 1 acceptable level; final mass in kg).
Run on the GPU box: python tools/ipm_delta3.py [K] [Nk] [max_iter] [nested -1|0|1] [key=value solver options, eng:key=value engine options ...]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lpopc_amd import problems
from lpopc_amd.engine import BatchedIPM, NLPEngine
from lpopc_amd.problem import Options

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
Nk = int(sys.argv[2]) if len(sys.argv) > 2 else 8
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
nested = int(sys.argv[4]) if len(sys.argv) > 4 else -1
extra = dict((kv.split("=")[0], float(kv.split("=")[1])) for kv in sys.argv[5:] if not kv.startswith("eng:"))
eng_extra = dict((kv[4:].split("=")[0], float(kv.split("=")[1])) for kv in sys.argv[5:] if kv.startswith("eng:"))
M_SCALE = 301454.0      # kg: total lift-off mass, the example's mass unit (example/launch/Launch.cpp)
o = Options()
o.SetStringValue("hessian-approximation", "exact")
eng = NLPEngine(problems.launch(K, Nk), o, device=0)
eng.set_option("ipm_nested", nested)
for k, v in eng_extra.items():
    eng.set_option(k, v)
ipm = BatchedIPM(eng, max_iter=iters, trace=iters, **extra)
x0 = eng.get_starting_point()[None, :]
if os.environ.get("IPM_PERTURB_SEED"):     # the path is chaotic: a start perturbed by 1e-10 (relative) is another sample of it
    import numpy as np
    x0 = x0 * (1 + 1e-10 * np.random.RandomState(int(os.environ["IPM_PERTURB_SEED"])).uniform(-1, 1, x0.shape))
ipm.solve(x0)           # warm-up: module load
t0 = time.perf_counter()
r = ipm.solve(x0)
dt = time.perf_counter() - t0
st, info, kt = ipm.stats(), ipm.info(), ipm.kernel_times()
tr = ipm.trace(0)
print(json.dumps({"problem": "Delta-III 4 x %d x %d" % (K, Nk), "n": eng.n, "m": eng.m, "kkt_order": info["kkt_order"],
                  "sub_problems": int(ipm.subproblems().shape[0]), "status": int(r["status"][0]), "iterations": int(r["iterations"][0]),
                  "restorations": int(ipm.restorations()[0]), "objective": float(r["obj"][0]), "final_mass_kg": -float(r["obj"][0]) * M_SCALE,
                  "kkt_error": float(r["kkt_error"][0]), "solve_s": dt, "ms_per_iteration": 1e3 * dt / max(1, st["iterations"]),
                  "factorizations": st["factorizations"], "trial_points": st["trial_points"], "factor_ms_total": kt["factor_ms"],
                  "substitution_ms_total": kt["substitution_ms"], "restoration_iterations": int((tr[:, 7] < 0).sum())}))
if os.environ.get("IPM_TRACE"):
    for i, t in enumerate(tr):
        print("%4d f=%.9f th=%.3e mu=%.1e a=%.2e az=%.2e dw=%.1e e0=%.3e ls=%d" % (i, *t[:7], int(t[7])))

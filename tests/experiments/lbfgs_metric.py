"""The metric problem with lpopc's default option hessian-approximation = limited-memory on the device.
python tests/experiments/lbfgs_metric.py [perturbation seed, 0 = lpopc's guess itself] [solver option=value ...]"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from lpopc_amd import problems
from lpopc_amd.engine import NLPEngine, BatchedIPM

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
e = NLPEngine(problems.launch(64, 16), device=0)
s = BatchedIPM(e, max_iter=3000, **dict((kv.split("=")[0], float(kv.split("=")[1])) for kv in sys.argv[2:]))
x0 = e.get_starting_point()[None, :]
if seed:
    x0 = x0 * (1 + 1e-10 * np.random.RandomState(seed).uniform(-1, 1, x0.shape))
t = time.time()
r = s.solve(x0)
dt = time.time() - t
st = s.stats()
print("status", int(r["status"][0]), "iterations", int(r["iterations"][0]), "mass", -float(r["obj"][0]) * 301454.0, "%.2f s" % dt, st,
      "%.2f ms/iter" % (1e3 * dt / max(1, st["iterations"])), s.info(), s.kernel_times(), flush=True)

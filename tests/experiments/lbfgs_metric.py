"""The metric problem with lpopc's default option hessian-approximation = limited-memory on the device (for a kernel trace)."""
import sys, time
sys.path.insert(0, ".")
from lpopc_amd import problems
from lpopc_amd.engine import NLPEngine, BatchedIPM

e = NLPEngine(problems.launch(64, 16), device=0)
s = BatchedIPM(e, max_iter=3000)
x0 = e.get_starting_point()[None, :]
t = time.time()
r = s.solve(x0)
dt = time.time() - t
st = s.stats()
print("status", int(r["status"][0]), "iterations", int(r["iterations"][0]), "mass", -float(r["obj"][0]) * 301454.0, "%.2f s" % dt, st,
      "%.2f ms/iter" % (1e3 * dt / max(1, st["iterations"])), s.info(), s.kernel_times(), flush=True)

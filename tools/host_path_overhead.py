import sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lpopc_amd import problems
from lpopc_amd.engine import NLPEngine
for name, mk in (("brachistochrone 1x10", lambda: problems.brachistochrone(1, 10)), ("launch 64x16", lambda: problems.config("launch"))):
    e = NLPEngine(mk(), device=0)
    e.set_option("pin_host", 1); e.set_option("const_once", 1)
    xl, xu, _, _ = e.get_bounds_info()
    x = problems.seeded_iterate(e.get_starting_point(), xl, xu, 1)
    g, v = np.zeros(e.m), np.zeros(e.nnz_jac)
    for _ in range(20):
        e.eval_g(x, True, out=g); e.eval_jac_g(x, False, out=v)
    for chk in (1, 0):
        e.set_option("check_finite", chk)
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < 1.0:
            e.eval_g(x, True, out=g); e.eval_jac_g(x, False, out=v); n += 1
        print("%-22s check_finite=%d: %.1f us per pair" % (name, chk, (time.perf_counter() - t0) / n * 1e6))
    e.close()

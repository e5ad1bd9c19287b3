"""Prints how closely the GPU exact Hessian matches the CPU oracle per problem (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lpopc_amd import problems
from lpopc_amd.engine import NLPEngine
from lpopc_amd.problem import Options
from oracle.oracle import Oracle
import time
o = Options(); o.SetStringValue("hessian-approximation", "exact")
for name, mk in [("bryson_denham", lambda: problems.bryson_denham(3, 5)), ("hypersensitive", lambda: problems.hypersensitive([-1, -0.5, 0.4, 1], [4, 6, 3], tf=50.0)),
                 ("brachistochrone", lambda: problems.brachistochrone(2, 6)), ("quadrotor", lambda: problems.quadrotor(2, 4)),
                 ("climb", lambda: problems.min_time_climb(2, 6)), ("launch 2x5", lambda: problems.launch(2, 5)), ("launch 64x16", lambda: problems.launch(64, 16))]:
    p = mk(); e = NLPEngine(p, o, device=0); r = Oracle(p, o)
    xl, xu, _, _ = e.get_bounds_info(); x = problems.seeded_iterate(e.get_starting_point(), xl, xu, 5)
    lam = np.random.RandomState(1).uniform(-1, 1, e.m)
    t = time.perf_counter(); hr = r.eval_h(x, 0.7, lam); tc = time.perf_counter() - t
    hv = e.eval_h(x, 0.7, lam); t = time.perf_counter(); hv = e.eval_h(x, 0.7, lam); tg = time.perf_counter() - t
    d = np.abs(hv - hr)
    print("%-16s nnz_h %7d  max|ref| %.3g  max abs diff %.3g  bit-identical entries %.1f%%  cpu %.1f ms gpu(host path) %.2f ms"
          % (name, e.nnz_h, np.abs(hr).max(), d.max(), 100.0 * np.mean(hv == hr), tc * 1e3, tg * 1e3))

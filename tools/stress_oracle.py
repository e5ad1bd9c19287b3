"""Randomised GPU-vs-oracle check on random ragged meshes (eval_f, eval_grad_f, eval_g, eval_jac_g, structure, bounds,
starting point) with the tolerances of tests/test_gpu_parity.py.  Run on the GPU box: python tools/stress_oracle.py [n] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from lpopc_amd import problems
from lpopc_amd.engine import NLPEngine
from oracle.oracle import Oracle

rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 30
makers = [("launch", problems.launch), ("bryson_denham", problems.bryson_denham), ("quadrotor", lambda: problems.quadrotor(2, 4)),
          ("climb", lambda: problems.min_time_climb(2, 4)), ("hypersensitive", lambda: problems.hypersensitive()),
          ("brachistochrone", lambda: problems.brachistochrone(1, 5))]


def rel(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))) if a.size else 0.0


worst = {"g": 0.0, "jac": 0.0, "f": 0.0, "grad": 0.0}
bad = 0
for t in range(trials):
    name, mk = makers[rng.integers(len(makers))]
    prob = mk()
    for i in range(prob.GetPhaseNum()):
        K = int(rng.integers(1, 6))
        ph = prob.GetPhase(i)
        ph.meshpoints = [-1.0] + [float(c) for c in np.sort(rng.uniform(-0.95, 0.95, K - 1))] + [1.0]
        ph.nodesperinterval = [int(v) for v in rng.integers(2, 25, K)]
    eng, orc = NLPEngine(prob, device=0), Oracle(prob)
    ok = (eng.n, eng.m, eng.nnz_jac) == (orc.n, orc.m, orc.nnz_jac)
    ok = ok and all(np.array_equal(a, b) for a, b in zip(eng.get_bounds_info(), orc.bounds()))
    ok = ok and np.array_equal(eng.get_starting_point(), orc.starting_point())
    ok = ok and all(np.array_equal(a, b) for a, b in zip(eng.eval_jac_g_structure(), orc.jac_structure()))
    xl, xu, _, _ = eng.get_bounds_info()
    x = problems.seeded_iterate(eng.get_starting_point(), xl, xu, t)
    e = {"g": rel(eng.eval_g(x), orc.eval_g(x)), "jac": rel(eng.eval_jac_g(x, False), orc.eval_jac_g(x)),
         "f": rel(np.ravel(eng.eval_f(x)), np.array([orc.eval_f(x)])), "grad": rel(eng.eval_grad_f(x), orc.eval_grad_f(x))}
    ok = ok and e["g"] <= 1e-12 and e["f"] <= 1e-12 and e["jac"] <= 1e-8 and e["grad"] <= 1e-8
    for k in worst:
        worst[k] = max(worst[k], e[k])
    if not ok:
        bad += 1
        print("MISMATCH", t, name, e, [prob.GetPhase(i).nodesperinterval for i in range(prob.GetPhaseNum())])
    eng.close()
print("worst relative errors", worst)
print("FAILED" if bad else "ALL WITHIN TOLERANCE", bad, "of", trials)
sys.exit(1 if bad else 0)

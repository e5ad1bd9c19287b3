import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from lpopc_amd import problems
from lpopc_amd.problem import Options
from lpopc_amd.engine import BatchedIPM, NLPEngine
o = Options(); o.SetStringValue("hessian-approximation", "exact")
prob = problems.launch()
eng = NLPEngine(prob, o, device=0)
ipm = BatchedIPM(eng, tol=1e-6)
r = ipm.solve(eng.get_starting_point()[None, :])
print("status", r["status"], "obj", r["obj"])
eng.finalize_solution(0, r["x"][0], r["lambda"][0], float(r["obj"][0]))
for i in range(4):
    ph = prob.GetPhase(i)
    print("phase", i, "mesh", ph.GetMeshPoints(), ph.GetNodesPerInterval())
    for tol, nmin, nmax in ((1e-3, 4, 10), (1e-6, 4, 10), (1e-6, 4, 30)):
        done, mesh, nodes, err = eng.ph_refine_mesh(i, tol, nmin, nmax)
        print("   tol", tol, "Nmin", nmin, "Nmax", nmax, "-> done", done, "mesh", np.round(mesh, 4).tolist(), "nodes", nodes, "err", err if np.isscalar(err) else np.max(err))

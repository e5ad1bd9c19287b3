import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from lpopc_amd import problems
from lpopc_amd.problem import Options
from oracle.oracle import Oracle
import ipm_x
from oracle import ipm_oracle
prob_name, K, Nk = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
opts = json.loads(sys.argv[4]) if len(sys.argv) > 4 else {}
o = Options(); o.SetStringValue("hessian-approximation", "exact")
mk = {"launch": lambda: problems.launch(K, Nk), "climb": lambda: problems.min_time_climb(K, Nk), "bd": lambda: problems.bryson_denham(K, Nk) if K else problems.bryson_denham(),
      "quad": lambda: problems.quadrotor(K, Nk), "brach": lambda: problems.brachistochrone(K, Nk)}[prob_name]
orc = Oracle(mk(), o)
t0 = time.time()
x0 = orc.starting_point()
seed = opts.pop('perturb_seed', 0)
if seed:
    x0 = x0 * (1 + 1e-10 * np.random.RandomState(seed).uniform(-1, 1, x0.size))
official = opts.pop('official', 0)
r = ipm_oracle.solve(orc, x0, **opts) if official else ipm_x.solve(orc, x0, **opts)
print("status", r["status"], "it", r["iterations"], "obj %.10f" % r["obj"], "err", r["kkt_error"], "resto", r["restorations"], "soc", sum(t.get("soc", 0) > 0 for t in r.get("trace", [])), "t %.1f" % (time.time() - t0))

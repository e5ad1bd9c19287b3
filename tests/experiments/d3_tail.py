import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from lpopc_amd import problems
from lpopc_amd.problem import Options
from oracle.oracle import Oracle
from oracle import ipm_oracle
K, Nk = int(sys.argv[1]), int(sys.argv[2])
opts = json.loads(sys.argv[3]) if len(sys.argv) > 3 else {}
o = Options(); o.SetStringValue("hessian-approximation", "exact")
orc = Oracle(problems.launch(K, Nk), o)
r = ipm_oracle.solve(orc, orc.starting_point(), **opts)
print("status", r["status"], r["iterations"], r["obj"], r["kkt_error"], r["restorations"])
for t in r["trace"][-25:]:
    print("%3d f=%.9f th=%.3e mu=%.1e a=%.2e az=%.2e dw=%.1e e0=%.3e ls=%d soc=%d dinf=%.1e cinf=%.1e comp=%.1e smin=%.1e" % (t["it"], t["f"], t["theta"], t["mu"], t["alpha"], t["alpha_z"], t["delta_w"], t["err0"], t["ls"], t.get("soc", 0), t.get("dinf",0), t.get("cinf",0), t.get("comp",0), t.get("smin",0)))
x = r["x"]; xl, xu, gl, gu = orc.bounds()
free = xl != xu
sl = np.where(free, x - xl, 1); su = np.where(free, xu - x, 1)
print("smallest slacks lo", np.sort(sl[free])[:6], "up", np.sort(su[free])[:6])

import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from lpopc_amd import problems
from lpopc_amd.problem import Options
from oracle.oracle import Oracle
from oracle import ipm_oracle
K, Nk = int(sys.argv[1]), int(sys.argv[2])
o = Options(); o.SetStringValue("hessian-approximation", "exact")
orc = Oracle(problems.launch(K, Nk), o)
x0 = orc.starting_point()
t0 = time.time()
r = ipm_oracle.solve(orc, x0, max_iter=int(sys.argv[3]))
print("status", r["status"], "it", r["iterations"], "obj", r["obj"], "err", r["kkt_error"], "resto", r["restorations"], "t", time.time() - t0)
for t in r["trace"][:80]:
    print("%3d f=%.6f th=%.3e mu=%.1e a=%.2e az=%.2e dw=%.1e e0=%.2e ls=%d" % (t["it"], t["f"], t["theta"], t["mu"], t["alpha"], t["alpha_z"], t["delta_w"], t["err0"], t["ls"]))

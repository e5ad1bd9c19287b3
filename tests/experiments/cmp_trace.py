import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from lpopc_amd import problems
from lpopc_amd.problem import Options
from lpopc_amd.engine import BatchedIPM, NLPEngine
from oracle.oracle import Oracle
from oracle import ipm_oracle
name = sys.argv[1]; tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-8
mk = {"bd": lambda: problems.bryson_denham(), "brach": lambda: problems.brachistochrone(1, 10), "bd28": lambda: problems.bryson_denham(2, 8),
      "launch26": lambda: problems.launch(2, 6), "climb": lambda: problems.min_time_climb(4, 6)}[name]
o = Options(); o.SetStringValue("hessian-approximation", "exact")
prob = mk()
orc = Oracle(prob, o)
ref = ipm_oracle.solve(orc, orc.starting_point(), tol=tol)
eng = NLPEngine(prob, o, device=0)
ipm = BatchedIPM(eng, tol=tol, trace=500)
r = ipm.solve(orc.starting_point()[None, :])
tr = ipm.trace(0)
print("oracle: status", ref["status"], "it", ref["iterations"], "obj", ref["obj"], "resto", ref["restorations"])
print("device: status", r["status"][0], "it", r["iterations"][0], "obj", r["obj"][0], "resto", ipm.restorations()[0])
for k in range(max(len(tr), len(ref["trace"]))):
    a = ref["trace"][k] if k < len(ref["trace"]) else None
    b = tr[k] if k < len(tr) else None
    sa = "O f=%.8f th=%.2e mu=%.1e a=%.2e az=%.2e dw=%.1e ls=%d soc=%d" % (a["f"], a["theta"], a["mu"], a["alpha"], a["alpha_z"], a["delta_w"], a["ls"], a.get("soc", 0)) if a else "O -"
    sb = "D f=%.8f th=%.2e mu=%.1e a=%.2e az=%.2e dw=%.1e ls=%d" % (b[0], b[1], b[2], b[3], b[4], b[5], int(b[7])) if b is not None else "D -"
    print("%3d %s | %s" % (k, sa, sb))

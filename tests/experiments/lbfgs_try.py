"""Device limited-memory solve vs the restatement on a few problems (experiment / smoke)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from lpopc_amd import problems
from lpopc_amd.engine import NLPEngine, BatchedIPM
from oracle import oracle as orc, ipm_oracle

for name, prob in (("param_sled", problems.param_sled(2, 12)), ("brach", problems.brachistochrone(2, 10)), ("bryson_denham", problems.bryson_denham(10, 4)),
                   ("param_osc", problems.param_oscillator()), ("quadrotor", problems.quadrotor(2, 4)), ("launch_2x6", problems.launch(2, 6))):
    e = NLPEngine(prob, device=0)
    s = BatchedIPM(e, max_iter=1500, trace=64)
    x0 = e.get_starting_point()[None, :]
    t = time.time()
    r = s.solve(x0)
    dt = time.time() - t
    o = orc.Oracle(prob)
    ro = ipm_oracle.solve(o, o.starting_point(), hessian_approximation="limited-memory", max_iter=1500) if name != "launch_2x6" else None
    print(name, "device:", int(r["status"][0]), int(r["iterations"][0]), float(r["obj"][0]), float(r["kkt_error"][0]), "%.2fs" % dt, s.stats(),
          "| oracle:", (ro["status"], ro["iterations"], ro["obj"]) if ro else None)
    if ro:
        tr = s.trace(0, 64)
        k = min(6, len(tr), len(ro["trace"]))
        for i in range(k):
            print("   it %d dev f=%.10g th=%.3e a=%.4g | orc f=%.10g th=%.3e a=%.4g" % (i, tr[i][0], tr[i][1], tr[i][3], ro["trace"][i]["f"], ro["trace"][i]["theta"], ro["trace"][i]["alpha"]))
    s.close(); e.close()

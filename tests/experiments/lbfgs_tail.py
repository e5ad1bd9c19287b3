import sys
sys.path.insert(0, ".")
import numpy as np
from lpopc_amd import problems
from lpopc_amd.engine import NLPEngine, BatchedIPM
e = NLPEngine(problems.launch(64, 16), device=0)
s = BatchedIPM(e, max_iter=3000, trace=3000)
x0 = e.get_starting_point()[None, :]
r = s.solve(x0)
tr = s.trace(0)
print("status", int(r["status"][0]), "iterations", int(r["iterations"][0]), len(tr), "restorations", int(s.restorations()[0]))
for i, t in list(enumerate(tr))[-30:]:
    print("%4d f=%.9f th=%.3e mu=%.1e a=%.2e az=%.2e dw=%.1e e0=%.3e ls=%d" % (i, *t[:7], int(t[7])))

"""Geometry of the metric problem's factorisation sub-problems (order, band part, border, half bandwidth, column stride), distinct rows with counts."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from lpopc_amd import problems
from lpopc_amd.engine import BatchedIPM, NLPEngine
from lpopc_amd.problem import Options

K = int(sys.argv[1]) if len(sys.argv) > 1 else 64
Nk = int(sys.argv[2]) if len(sys.argv) > 2 else 16
o = Options()
o.SetStringValue("hessian-approximation", "exact")
eng = NLPEngine(problems.launch(K, Nk), o, device=0)
eng.set_option("ipm_nested", -1)
for kv in sys.argv[3:]:
    eng.set_option(kv.split("=")[0], float(kv.split("=")[1]))
ipm = BatchedIPM(eng)
g = ipm.subproblems()
rows, first, counts = np.unique(g, axis=0, return_index=True, return_counts=True)
for i in np.argsort(first):
    print("first at %4d  x %3d  Nt %4d Nb %4d nb %3d b %3d CS %3d" % (first[i], counts[i], *rows[i]))
print(ipm.info())

"""Geometry of the KKT factorisation's sub-problems for a problem / mesh (no GPU work beyond rpm_ipm_create).
python tools/ipm_subproblems.py launch|quadrotor K Nk"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lpopc_amd import problems
from lpopc_amd.engine import BatchedIPM, NLPEngine
from lpopc_amd.problem import Options
name, K, Nk = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
o = Options(); o.SetStringValue("hessian-approximation", "exact")
eng = NLPEngine(getattr(problems, name)(K, Nk), o, device=0)
ipm = BatchedIPM(eng)
sp = ipm.subproblems()
print("sub-problems", sp.shape[0], "info", ipm.info())
seen = {}
for g in map(tuple, sp):
    seen[g] = seen.get(g, 0) + 1
for g, c in seen.items():
    print("  %4d x  order %4d  band %4d  border %3d  half bandwidth %3d  CS %3d   rows per block column %d" % (c, *g, 16 + g[3] + g[2]))

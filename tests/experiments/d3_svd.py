import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from lpopc_amd import problems
from lpopc_amd.problem import Options
from oracle.oracle import Oracle
import ipm_x
from oracle.ipm_oracle import DEFAULTS, _push
o = Options(); o.SetStringValue("hessian-approximation", "exact")
orc = Oracle(problems.launch(2, 6), o)
oo = dict(DEFAULTS); oo.update(ipm_x.X)
p = ipm_x.P(orc, oo)
x0 = orc.starting_point()
x = np.where(p.xl == p.xu, p.xl, _push(x0, p.xl, p.xu, oo))
v = np.concatenate([x, _push(orc.eval_g(x)[p.ineq], p.vl[p.n:], p.vu[p.n:], oo)])
A = p.jac(v)[:, p.free]
sv = np.linalg.svd(A, compute_uv=False)
print("m", p.m, "nv free", p.free.sum(), "fixed", (~p.free).sum(), "ineq", p.ns)
print("sv max %.3e min %.3e" % (sv[0], sv[-1]), sv[-8:])
c = p.cons(v)
print("theta", np.abs(c).sum(), "max c", np.abs(c).max(), "argmax", np.argmax(np.abs(c)))
print("bounds: nlo", p.lo.sum(), "nup", p.up.sum(), "min width", (p.vu - p.vl)[p.lo & p.up].min())
g = p.grad(v); print("grad nz", np.nonzero(g)[0], g[np.nonzero(g)[0]])
xl, xu = p.xl, p.xu
print("x range", x.min(), x.max())
# names of rows with largest violation
idx = np.argsort(-np.abs(c))[:20]
print(idx, c[idx])
U, S, Vt = np.linalg.svd(p.jac(v)[:, p.free])
for k in range(1, 4):
    u = U[:, -k]
    rows = np.nonzero(np.abs(u) > 1e-6)[0]
    print("left null", k, rows, u[rows].round(3))
Afull = p.jac(v)
zr = np.nonzero(np.abs(Afull).sum(1) == 0)[0]
print("zero rows", zr, "gl", p.gl[zr], "c", c[zr])
print("m per phase?", orc.m)

"""Ipopt's gradient-based NLP scaling (nlp_scaling_method = gradient-based, max gradient 100) around the restatement: iteration
counts with and without, on Delta-III meshes and the small problems (experiment)."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from lpopc_amd import problems
from lpopc_amd.problem import Options
from oracle import oracle as orc, ipm_oracle


class Scaled:
    def __init__(self, o, x0, gmax=100.0):
        self.o, self.n, self.m = o, o.n, o.m
        gr = np.abs(o.eval_grad_f(x0)).max()
        self.sf = gmax / gr if gr > gmax else 1.0
        ji, jj = o.jac_structure()
        jv = np.abs(o.eval_jac_g(x0))
        rmax = np.zeros(o.m)
        np.maximum.at(rmax, ji, jv)
        self.sc = np.where(rmax > gmax, gmax / np.maximum(rmax, 1e-300), 1.0)
        self.ji = ji
    def bounds(self):
        xl, xu, gl, gu = self.o.bounds()
        big = 1e19
        return xl, xu, np.where(np.abs(gl) < big, gl * self.sc, gl), np.where(np.abs(gu) < big, gu * self.sc, gu)
    def starting_point(self): return self.o.starting_point()
    def jac_structure(self): return self.o.jac_structure()
    def hess_structure(self): return self.o.hess_structure()
    def eval_f(self, x): return self.sf * self.o.eval_f(x)
    def eval_grad_f(self, x): return self.sf * self.o.eval_grad_f(x)
    def eval_g(self, x): return self.sc * self.o.eval_g(x)
    def eval_jac_g(self, x): return self.sc[self.ji] * self.o.eval_jac_g(x)
    def eval_h(self, x, s, lam): return self.o.eval_h(x, s * self.sf, lam * self.sc)


o = Options(); o.SetStringValue("hessian-approximation", "exact")
cases = [("launch_2x6", problems.launch(2, 6)), ("launch_4x8", problems.launch(4, 8)), ("launch_8x8", problems.launch(8, 8)),
         ("brach", problems.brachistochrone(2, 10)), ("bd", problems.bryson_denham(2, 8)), ("quadrotor_3x6", problems.quadrotor(3, 6))]
if len(sys.argv) > 1:
    cases = [c for c in cases if c[0] in sys.argv[1:]]
for name, prob in cases:
    O = orc.Oracle(prob, o)
    x0 = O.starting_point()
    for scaled in (0, 1):
        P = Scaled(O, x0) if scaled else O
        t = time.time()
        r = ipm_oracle.solve(P, x0, max_iter=1500)
        obj = r["obj"] / (P.sf if scaled else 1.0)
        g = O.eval_g(r["x"]); xl, xu, gl, gu = O.bounds()
        viol = max((gl - g).max(), (g - gu).max(), 0.0)
        extra = "sf %.3g sc min %.3g, %d rows scaled" % (P.sf, P.sc.min(), (P.sc < 1).sum()) if scaled else ""
        print(name, "scaled" if scaled else "plain ", "status", r["status"], "it", r["iterations"], "obj %.10g" % obj, "viol %.2e" % viol, "%.1fs" % (time.time() - t), extra, flush=True)

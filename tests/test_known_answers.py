"""Known-answer tests: optimal-control problems with analytic optima, solved through the NLP callbacks.

The reference ships no golden vectors, so these are the closest thing to an external pin: the callbacks (objective,
gradient, constraints, Jacobian, bounds, starting point) of the transcribed problem, handed to a generic NLP solver
(scipy's trust-constr; Ipopt is not in this image), must reproduce optima known in closed form.
  * Bryson-Denham (state-constrained double integrator, l = 1/9): J* = 4 / (9 l) = 4.
  * Brachistochrone: T* from the cycloid through the end point.
  * Hypersensitive (Lagrange cost, turnpike): for tf >> 1 the optimum is V(x0) + W(xf) with V' = -x^3 + sqrt(x^6 + x^2)
    (decay to the origin) and W' = x^3 + sqrt(x^6 + x^2) (arrival from it), from the Hamilton-Jacobi-Bellman equation
    of  min 1/2 int (x^2 + u^2),  x' = -x^3 + u.
CPU: the oracle's callbacks.  GPU: the product's, inside the reference's outer loop (solve -> extract -> estimate ->
refine) through the LpopcApplication mirror."""
import numpy as np
import pytest

from lpopc_amd import problems
from oracle import oracle as orc


class _OracleNLP:
    """The oracle behind the method names ScipyNLPSolver drives."""

    def __init__(self, o):
        self.o, self.n, self.m = o, o.n, o.m
        self.sol = None

    def get_bounds_info(self):
        return self.o.bounds()

    def get_starting_point(self):
        return self.o.starting_point()

    def eval_jac_g_structure(self):
        return self.o.jac_structure()

    def eval_f(self, x):
        return self.o.eval_f(x)

    def eval_grad_f(self, x):
        return self.o.eval_grad_f(x)

    def eval_g(self, x):
        return self.o.eval_g(x)

    def eval_jac_g(self, x):
        return self.o.eval_jac_g(x)

    def finalize_solution(self, status, x, lam, obj):
        self.sol = (np.array(x), np.array(lam), obj)


def _cycloid_time(xf, yf, g):
    """Minimum descent time from rest at the origin to (xf, yf), y measured downwards."""
    from scipy.optimize import brentq
    th = brentq(lambda t: (t - np.sin(t)) / (1 - np.cos(t)) - xf / yf, 1e-6, 2 * np.pi - 1e-6)
    R = yf / (1 - np.cos(th))
    return th * np.sqrt(R / g)


def test_bryson_denham_optimum_with_oracle_callbacks():
    from lpopc_amd.application import ScipyNLPSolver
    nlp = _OracleNLP(orc.Oracle(problems.bryson_denham()))
    assert ScipyNLPSolver(1e-6).SolveNlp(nlp)
    x, _, obj = nlp.sol
    assert abs(obj - 4.0) < 2e-2                     # 20 LGR points on one interval across two junctions
    M = 21
    assert x[:M].max() <= 1.0 / 9.0 + 1e-8 and x[:M].max() > 1.0 / 9.0 - 1e-4   # the state rides its bound
    xl, xu, gl, gu = nlp.o.bounds()
    g = nlp.o.eval_g(x)
    assert max((gl - g).max(), (g - gu).max()) < 1e-6


def test_brachistochrone_optimum_with_oracle_callbacks():
    from lpopc_amd.application import ScipyNLPSolver
    prob = problems.brachistochrone(2, 10)
    o = orc.Oracle(prob)
    nlp = _OracleNLP(o)
    assert ScipyNLPSolver(1e-6).SolveNlp(nlp)
    x, _, obj = nlp.sol
    ev = prob.GetPhase(0).GeteventMin()              # x0, y0, v0, xf, yf
    T = _cycloid_time(ev[3], ev[4], prob.GetOpimalProblemFuns().consts[0])
    assert abs(obj - T) < 1e-4 * T, (obj, T)
    assert abs(x[-1] - T) < 1e-4 * T


def test_hypersensitive_turnpike_cost_with_oracle_callbacks():
    """The Lagrange-cost path (quadrature, gradient of the integrand, analytic first derivatives): J* = V(1.5) + W(1)."""
    from scipy.integrate import quad
    from lpopc_amd.application import ScipyNLPSolver
    from lpopc_amd.problem import Options
    V = quad(lambda x: -x ** 3 + np.sqrt(x ** 6 + x ** 2), 0, 1.5)[0]
    W = quad(lambda x: x ** 3 + np.sqrt(x ** 6 + x ** 2), 0, 1.0)[0]
    K, n = 6, 10
    prob = problems.hypersensitive(np.linspace(-1, 1, K + 1).tolist(), [n] * K, tf=30.0)
    opts = Options()
    opts.SetStringValue("first-derive", "analytic")
    nlp = _OracleNLP(orc.Oracle(prob, opts))
    assert ScipyNLPSolver(1e-7, maxiter=300).SolveNlp(nlp)
    assert abs(nlp.sol[2] - (V + W)) < 1e-3 * (V + W), (nlp.sol[2], V + W)


@pytest.mark.gpu
def test_application_loop_on_gpu_reaches_the_analytic_optimum(built, tmp_path):
    """LpopcApplication mirror: NLP solve (scipy stand-in) -> Nlp2OpControl -> error estimate -> ph refinement ->
    next mesh ..., every callback and post-solve step on the GPU.  The solution sits at 4 / (9 l) on every mesh and the
    meshes grow where the estimate asks; like the reference, the loop ends with NoMoreRefine or with the max-grid error."""
    from lpopc_amd.application import LpopcApplication, console_not_print
    from lpopc_amd.problem import LpopcException
    prob = problems.bryson_denham(2, 8)
    app = LpopcApplication(console_not_print)
    app.SetOptimalControlProblem(prob)
    app.Options().SetNumericValue("desired-relative-error", 1e-7)
    app.Options().SetIntegerValue("max-grid-num", 3)
    app.Options().SetIntegerValue("Nmax", 12)
    try:
        from lpopc_amd.application import ScipyNLPSolver   # the stand-in this test is about (the default is the device solver now)
        finished = app.SolveOptimalProblem(nlp_solver=ScipyNLPSolver(app.Options().GetNumericValue("Ipopt-tol")), result_dir=tmp_path)
    except LpopcException as e:
        finished = False
        assert "max number of refine grid" in str(e)
    assert app.meshrefiner_.CurrentGrid() >= 1                      # the first mesh was not good enough
    assert abs(app.objective - 4.0) < 1e-3
    nodes = prob.GetPhase(0).GetNodesPerInterval()
    assert sum(nodes) > 16 and len(app.meshrefiner_.meshhistory) >= 2
    r = app.result[0]
    M = r["time"].size
    assert r["state"][:M].max() <= 1.0 / 9.0 + 1e-6
    if finished:
        assert (tmp_path / "state1").exists() and (tmp_path / "Hamiltonian1").exists()


@pytest.mark.gpu
def test_application_loop_with_hp_liu_refinement(built):
    """The same loop with mesh-refine-methods=hp-Liu (LiuHpMeshRefineAlg behind rpm_hpliu_*): +3 nodes on the first
    pass, divisions / reductions / merges afterwards; the objective stays at 4 / (9 l) and the loop ends like the
    reference's (NoMoreRefine, the max-grid error, or the exception the reference would raise from its history look-ups)."""
    from lpopc_amd.application import LpopcApplication, console_not_print
    from lpopc_amd.engine import RpmError
    from lpopc_amd.problem import LpopcException
    prob = problems.bryson_denham(2, 8)
    app = LpopcApplication(console_not_print)
    app.SetOptimalControlProblem(prob)
    app.Options().SetStringValue("mesh-refine-methods", "hp-Liu")
    app.Options().SetNumericValue("desired-relative-error", 1e-6)
    app.Options().SetIntegerValue("max-grid-num", 4)
    app.Options().SetIntegerValue("Nmax", 12)
    try:
        from lpopc_amd.application import ScipyNLPSolver   # the stand-in this test was written around (the default is the device solver now)
        app.SolveOptimalProblem(nlp_solver=ScipyNLPSolver(app.Options().GetNumericValue("Ipopt-tol")))
    except (LpopcException, RpmError):
        pass
    assert app.meshrefiner_.CurrentGrid() >= 1 and abs(app.objective - 4.0) < 1e-3
    first, second = app.meshrefiner_.meshhistory[0][0], app.meshrefiner_.meshhistory[1][0]
    assert first.nodesPerInterval == [8, 8]
    assert all(n in (8, 11) or 2 <= n <= 8 for n in second.nodesPerInterval)   # +3 where unsatisfied, reduced where satisfied


@pytest.mark.gpu
def test_analytic_derive_check_option(built):
    """analytic-derive-check=yes: the hypersensitive problem's analytic derivatives against forward differences at the
    guess; a deliberately huge tolerance-perturbation makes the finite differences disagree grossly."""
    from lpopc_amd.application import LpopcApplication, console_not_print
    prob = problems.hypersensitive(np.linspace(-1, 1, 5).tolist(), [6] * 4, tf=30.0)
    app = LpopcApplication(console_not_print)
    app.SetOptimalControlProblem(prob)
    app.Options().SetStringValue("first-derive", "analytic")
    app.Options().SetStringValue("analytic-derive-check", "yes")
    app.Options().SetNumericValue("analytic-derive-check-tol", 1e-5)
    small = app.CheckAnalyticDerive()
    # like the reference's checker the same number is perturbation and threshold, so curved functions are always reported:
    # here by the forward-difference truncation error h f''/2 (times dt/2), nothing larger
    assert all(abs(a - f) < 1e-2 for _, _, a, f in small)
    app.Options().SetNumericValue("analytic-derive-check-tol", 1e-1)   # h ~ 0.1: forward differences of -x^3 are far off
    bad = app.CheckAnalyticDerive()
    assert bad and max(abs(a - f) for _, _, a, f in bad) > 1.0
    assert all(k[0] in ("jacobian value", "objective gradient") for k in bad)

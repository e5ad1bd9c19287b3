"""Static parameters (nq > 0; Phase(idx, nx, nu, nq, nc, ne) + SetparameterlMin / SetparameterMax / SetparameterGuess,
Core/LpOptimalProblem.hpp:33-187).  Layout, bounds, guess and the order of every derivative column / Jacobian block follow the
reference (Core/LpBoundsChecker.cpp:117-138, Core/LpFiniteDifferenceDerive.cpp:282-317, Core/LpNLPWrapper.cpp:763-769,
814-820, 854-859, 461-519, 1088-1097); the values follow the mathematically correct formulas where the reference's own
parameter path contradicts itself (oracle/orpm_core.c lists each departure) — so here the oracle
is the specification and the tests that need no oracle carry the weight: Jacobian against central differences of eval_g,
gradient against central differences of eval_f, analytic against finite-difference mode, a closed-form optimum.
Two problems authored for this: a minimum-time sled with one design parameter (optimum p = 1, cost 2.5) and a two-phase
oscillator whose every callback — dynamics, path, costs, events, linkage — depends on two parameters per phase."""
import numpy as np
import pytest
import scipy.sparse as sp

from lpopc_amd import problems
from lpopc_amd.engine import NLPEngine, RpmError
from lpopc_amd.problem import Options
from oracle import oracle as orc

G_TOL, JFD_TOL = 1e-12, 1e-8
PROBLEMS = [("param_sled", lambda: problems.param_sled(4, 8)), ("param_oscillator", lambda: problems.param_oscillator()),
            ("param_oscillator_ragged", lambda: _ragged())]


def _ragged():
    p = problems.param_oscillator()
    problems_set = [([-1, -0.7, 0.2, 1], [4, 19, 3]), ([-1, 0.1, 1], [17, 5])]
    for i, (mesh, nodes) in enumerate(problems_set):
        ph = p.GetPhase(i)
        ph.GetMeshPoints().clear()
        ph.GetNodesPerInterval().clear()
        problems.set_mesh(ph, mesh, nodes)
    return p


def rel_err(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))) if a.size else 0.0


def _iterate(o, seed):
    xl, xu, _, _ = o.bounds()
    return problems.seeded_iterate(o.starting_point(), xl, xu, seed)


@pytest.mark.parametrize("name,make", PROBLEMS, ids=[p[0] for p in PROBLEMS])
def test_layout_and_derivatives_without_a_device(built, name, make):
    prob = make()
    e, o = NLPEngine(prob), orc.Oracle(prob)
    # layout: [X | U | t0 tf | p] per phase; host set-up equals the oracle's bit for bit
    nq = [prob.GetPhase(i).get_optimal_info()[2] for i in range(prob.GetPhaseNum())]
    assert (e.n, e.m, e.nnz_jac) == (o.n, o.m, o.nnz_jac) and sum(nq) > 0
    for a, b in zip(e.get_bounds_info(), o.bounds()):
        assert np.array_equal(a, b)
    assert np.array_equal(e.get_starting_point(), o.starting_point())
    i, j = e.eval_jac_g_structure()
    oi, oj = o.jac_structure()
    assert np.array_equal(i, oi) and np.array_equal(j, oj)
    assert len(set(zip(i.tolist(), j.tolist()))) == i.size              # no duplicate entries (App. A.3)
    # the parameter columns are where the bounds / guess put the parameters
    xl, xu, _, _ = e.get_bounds_info()
    x0 = e.get_starting_point()
    off = 0
    for ip in range(prob.GetPhaseNum()):
        ph = prob.GetPhase(ip)
        nx, nu, q, _, _ = ph.get_optimal_info()
        N = sum(ph.GetNodesPerInterval())
        p0 = off + nx * (N + 1) + nu * N + 2
        assert np.array_equal(xl[p0:p0 + q], ph.GetparameterMin()) and np.array_equal(xu[p0:p0 + q], ph.GetparameterMax())
        assert np.array_equal(x0[p0:p0 + q], ph.GetparameterGuess())
        off = p0 + q
    assert off == e.n
    # oracle-free checks of the formulas: central differences of eval_g / eval_f
    x = _iterate(o, 3)
    J = sp.coo_matrix((o.eval_jac_g(x), (oi, oj)), shape=(o.m, o.n)).toarray()
    grad = o.eval_grad_f(x)
    Jc, gc = np.zeros_like(J), np.zeros(o.n)
    for k in range(o.n):
        h = 1e-6 * (1 + abs(x[k]))
        xp, xm = x.copy(), x.copy()
        xp[k] += h
        xm[k] -= h
        Jc[:, k] = (o.eval_g(xp) - o.eval_g(xm)) / (2 * h)
        gc[k] = (o.eval_f(xp) - o.eval_f(xm)) / (2 * h)
    assert np.max(np.abs(J - Jc)) < 2e-7 and np.max(np.abs(grad - gc)) < 5e-6
    # every parameter column of the Jacobian is really populated (the derivative is not silently zero)
    off = 0
    for ip in range(prob.GetPhaseNum()):
        ph = prob.GetPhase(ip)
        nx, nu, q, _, _ = ph.get_optimal_info()
        N = sum(ph.GetNodesPerInterval())
        p0 = off + nx * (N + 1) + nu * N + 2
        for c in range(q):
            assert np.abs(J[:, p0 + c]).max() > 1e-3
        off = p0 + q
    e.close()


def test_analytic_mode_equals_finite_differences_with_a_parameter(built):
    prob = problems.param_sled(3, 7)
    an = Options()
    an.SetStringValue("first-derive", "analytic")
    a, f = orc.Oracle(prob, an), orc.Oracle(prob)
    x = _iterate(f, 5)
    assert rel_err(a.eval_jac_g(x), f.eval_jac_g(x)) < 1e-6 and rel_err(a.eval_grad_f(x), f.eval_grad_f(x)) < 1e-5
    assert np.array_equal(a.eval_g(x), f.eval_g(x))


def test_sled_optimum_is_the_closed_form(built):
    """min tf + 0.5 p^2 with tf = 2 / sqrt(p) (bang-bang): p* = 1, J* = 2.5 — through the oracle's callbacks and scipy."""
    from lpopc_amd.application import ScipyNLPSolver
    from tests.test_known_answers import _OracleNLP
    o = orc.Oracle(problems.param_sled(2, 12))           # the switch of the bang-bang control falls on the mesh point
    nlp = _OracleNLP(o)
    assert ScipyNLPSolver(1e-9, maxiter=2000).SolveNlp(nlp)
    x, _, obj = nlp.sol
    assert abs(obj - 2.5) < 1e-5 and abs(x[-1] - 1.0) < 1e-4 and abs(x[-2] - 2.0) < 1e-4       # [.., t0, tf, p]


def test_parameter_set_up_errors(built):
    p = problems.param_sled()
    p.GetPhase(0).vparametermin[0] = 20.0                                     # min > max
    with pytest.raises(RpmError) as ei:
        NLPEngine(p)
    assert "Bounds on parameter are Inconsistent" in str(ei.value)
    from lpopc_amd.problem import OptimalProblem, ProblemFunctor
    q = problems.param_sled()
    wrong = OptimalProblem(1, 0, ProblemFunctor(problems.RPM_PROBLEM_BRYSON_DENHAM, []))   # a functor without parameters
    wrong.AddPhase(q.GetPhase(0))
    with pytest.raises(RpmError):
        NLPEngine(wrong)
    ex = Options()
    ex.SetStringValue("hessian-approximation", "exact")
    with pytest.raises(RpmError) as ei:
        NLPEngine(problems.param_sled(), ex)
    assert "static parameters" in str(ei.value)


# ---------------------------------------------------------------------------------------------------- on the GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name,make", PROBLEMS, ids=[p[0] for p in PROBLEMS])
@pytest.mark.parametrize("tile", [0, 16, 64])
def test_gpu_callbacks_against_the_oracle(built, name, make, tile):
    prob = make()
    eng, o = NLPEngine(prob, tile_nodes=tile, device=0), orc.Oracle(prob)
    for seed in (3, 11):
        x = _iterate(o, seed)
        g, v = eng.eval_g(x, True), eng.eval_jac_g(x, False)
        assert rel_err(g, o.eval_g(x)) <= G_TOL and rel_err(v, o.eval_jac_g(x)) <= JFD_TOL
        assert abs(eng.eval_f(x) - o.eval_f(x)) <= G_TOL * max(1.0, abs(o.eval_f(x)))
        assert rel_err(eng.eval_grad_f(x), o.eval_grad_f(x)) <= JFD_TOL
        g2, v2 = eng.eval_pair(x)
        assert np.array_equal(g2, g) and np.array_equal(v2, v)
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,make", PROBLEMS[1:], ids=[p[0] for p in PROBLEMS[1:]])
def test_gpu_layouts_batches_and_sharding_are_bit_identical(built, name, make):
    import torch
    from lpopc_amd.group import EngineGroup
    prob, B = make(), 37
    one = NLPEngine(prob, device=0)
    o = orc.Oracle(prob)
    xs = np.stack([_iterate(o, 40 + b) for b in range(B)])
    ref = [one.eval_pair(x) for x in xs]
    dx = torch.from_numpy(xs).cuda()
    for kw, pl in ((dict(role_loop=0), -1), (dict(role_loop=1), 0), (dict(role_loop=1), 1)):
        e = NLPEngine(prob, n_instances=B, device=0, **kw)
        if pl >= 0:
            e.set_option("pipeline", pl)
        dg = torch.full((B, e.m), np.nan, dtype=torch.float64, device="cuda")
        dv = torch.full((B, e.nnz_jac), np.nan, dtype=torch.float64, device="cuda")
        e.eval_pair_dev(dx, dg, dv)
        torch.cuda.synchronize()
        assert pl < 1 or e.get_option("pipeline_active") == 1
        g, v = dg.cpu().numpy(), dv.cpu().numpy()
        for b in range(B):
            assert np.array_equal(g[b], ref[b][0]) and np.array_equal(v[b], ref[b][1]), (kw, pl, b)
        e.close()
    # mesh intervals sharded over three engines (one process, the same device three times): the parameter blocks, the event
    # and linkage entries on the parameters land where a single engine puts them
    import mmap
    own = lambda n: np.frombuffer(mmap.mmap(-1, 8 * n), dtype=np.float64, count=n)   # noqa: E731
    grp = EngineGroup(prob, [0, 0, 0])
    x, g, v = own(one.n), own(one.m), own(one.nnz_jac)
    for b in (0, 5):
        x[:] = xs[b]
        grp.eval_pair(x, g, v)
        assert np.array_equal(g, ref[b][0]) and np.array_equal(v, ref[b][1])
    grp.close()
    one.close()


@pytest.mark.gpu
def test_gpu_analytic_mode_with_a_parameter(built):
    prob = problems.param_sled(5, 9)
    an = Options()
    an.SetStringValue("first-derive", "analytic")
    eng, o = NLPEngine(prob, an, device=0), orc.Oracle(prob, an)
    x = _iterate(o, 9)
    assert rel_err(eng.eval_g(x), o.eval_g(x)) <= G_TOL and rel_err(eng.eval_jac_g(x), o.eval_jac_g(x)) <= 1e-12
    assert rel_err(eng.eval_grad_f(x), o.eval_grad_f(x)) <= 1e-12
    eng.close()

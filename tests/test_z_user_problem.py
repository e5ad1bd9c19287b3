"""A user's problem without rebuilding the package's library (the stand-in for subclassing lpopc's FunctionWrapper,
Core/LpFunctionWrapper.h:50-69): lpopc_amd.userproblem.build() compiles the engine around a functor header.
CPU half: the build works here (hipcc cross-compiles), the library exports the whole C ABI and its host-side set-up equals
the built-in functor's.  GPU half: results bit-identical to the package's own library for functors that exist in both
(and thereby equal to the oracle), on every kernel layout; a problem that exists ONLY as a user header passes the
oracle-free derivative checks."""
import ctypes as C
import os

import numpy as np
import pytest

from lpopc_amd import problems, userproblem
from lpopc_amd.engine import ABI_SYMBOLS, NLPEngine, RpmError
from lpopc_amd.problem import OptimalProblem, Options, Phase, ProblemFunctor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _as_user(prob, header):
    f = prob.GetOpimalProblemFuns()
    f.problem_id = userproblem.RPM_PROBLEM_USER
    f.library = userproblem.build(os.path.join(ROOT, header))
    return prob


def test_user_library_builds_exports_the_abi_and_sets_up_like_the_builtin(built):
    so = userproblem.build(os.path.join(ROOT, "tests", "user_problems", "bryson_denham_user.hpp"))
    assert os.path.exists(so) and so == userproblem.build(os.path.join(ROOT, "tests", "user_problems", "bryson_denham_user.hpp"))   # cached
    L = C.CDLL(so)
    for sym in ABI_SYMBOLS:
        assert hasattr(L, sym), sym
    a = NLPEngine(problems.bryson_denham())
    b = NLPEngine(_as_user(problems.bryson_denham(), "tests/user_problems/bryson_denham_user.hpp"))
    assert (a.n, a.m, a.nnz_jac) == (b.n, b.m, b.nnz_jac)
    for u, v in zip(a.eval_jac_g_structure(), b.eval_jac_g_structure()):
        assert np.array_equal(u, v)
    for u, v in zip(a.get_bounds_info(), b.get_bounds_info()):
        assert np.array_equal(u, v)
    assert np.array_equal(a.get_starting_point(), b.get_starting_point())
    # the user library was built without the package's functors: their ids are refused, loudly
    p = problems.bryson_denham()
    p.GetOpimalProblemFuns().library = so
    with pytest.raises(RpmError):
        NLPEngine(p)
    # and the package's library does not know the user id
    q = problems.bryson_denham()
    q.GetOpimalProblemFuns().problem_id = userproblem.RPM_PROBLEM_USER
    with pytest.raises(RpmError):
        NLPEngine(q)


@pytest.mark.gpu
@pytest.mark.parametrize("name,make,header,analytic", [
    ("bryson_denham", lambda: problems.bryson_denham(), "tests/user_problems/bryson_denham_user.hpp", False),
    ("brachistochrone_fd", lambda: problems.brachistochrone(3, 7), "tests/user_problems/brachistochrone_user.hpp", False),
    ("brachistochrone_analytic", lambda: problems.brachistochrone(3, 7), "tests/user_problems/brachistochrone_user.hpp", True),
])
def test_user_library_is_bit_identical_to_the_builtin_functor(built, name, make, header, analytic):
    import torch
    from oracle.oracle import Oracle
    opts = Options()
    opts.SetStringValue("hessian-approximation", "exact")
    if analytic:
        opts.SetStringValue("first-derive", "analytic")
    a = NLPEngine(make(), opts, device=0)
    b = NLPEngine(_as_user(make(), header), opts, device=0)
    xl, xu, _, _ = a.get_bounds_info()
    x = problems.seeded_iterate(a.get_starting_point(), xl, xu, 3)
    lam = np.random.RandomState(2).uniform(-1, 1, a.m)
    assert np.array_equal(a.eval_g(x), b.eval_g(x)) and np.array_equal(a.eval_jac_g(x, False), b.eval_jac_g(x, False))
    assert a.eval_f(x) == b.eval_f(x) and np.array_equal(a.eval_grad_f(x), b.eval_grad_f(x))
    assert a.nnz_h == b.nnz_h and np.array_equal(a.eval_h(x, 0.7, lam), b.eval_h(x, 0.7, lam))
    orc = Oracle(make(), opts)
    assert np.max(np.abs(b.eval_g(x) - orc.eval_g(x)) / np.maximum(1, np.abs(orc.eval_g(x)))) <= 1e-12
    # the throughput layouts, batched
    B = 40
    xs = np.stack([problems.seeded_iterate(a.get_starting_point(), xl, xu, 10 + i) for i in range(B)])
    out = []
    for mk in (make, lambda: _as_user(make(), header)):
        for pipeline in (0, 1):
            e = NLPEngine(mk(), opts, n_instances=B, device=0, role_loop=1)
            e.set_option("pipeline", pipeline)
            dg = torch.empty((B, e.m), dtype=torch.float64, device="cuda")
            dv = torch.empty((B, e.nnz_jac), dtype=torch.float64, device="cuda")
            e.eval_pair_dev(torch.from_numpy(xs).cuda(), dg, dv)
            torch.cuda.synchronize()
            assert e.get_option("pipeline_active") == pipeline
            out.append((dg.cpu().numpy(), dv.cpu().numpy()))
            e.close()
    for g, v in out[1:]:
        assert np.array_equal(g, out[0][0]) and np.array_equal(v, out[0][1])
    a.close()
    b.close()


def _vanderpol(n_intervals=6, nodes=7):
    ph = Phase(1, 2, 1, 0, 0, 0)
    ph.SetTimeMin(0.0, 5.0)
    ph.SetTimeMax(0.0, 5.0)
    for lo, hi, x0 in ((-5.0, 5.0, 1.0), (-5.0, 5.0, 0.0)):
        ph.SetStateMin(x0, lo, lo)
        ph.SetStateMax(x0, hi, hi)
    ph.SetcontrolMin(-0.3)
    ph.SetcontrolMax(1.0)
    ph.SetTimeGuess(0.0)
    ph.SetTimeGuess(5.0)
    ph.SetStateGuess(1, 1.0)
    ph.SetStateGuess(1, 0.0)
    ph.SetStateGuess(2, 0.0)
    ph.SetStateGuess(2, 0.0)
    ph.SetControlGuess(1, 0.0)
    ph.SetControlGuess(1, 0.0)
    problems.set_mesh(ph, np.linspace(-1, 1, n_intervals + 1), [nodes] * n_intervals)
    lib = userproblem.build(os.path.join(ROOT, "examples", "user_problem_vanderpol.hpp"))
    op = OptimalProblem(1, 0, ProblemFunctor(userproblem.RPM_PROBLEM_USER, [1.0], library=lib))
    op.AddPhase(ph)
    return op


@pytest.mark.gpu
def test_a_problem_that_exists_only_as_a_user_header(built):
    """examples/user_problem_vanderpol.hpp: no oracle function exists for it, so the oracle-free checks of SURVEY §4: the
    Jacobian against central differences of eval_g, the gradient against central differences of eval_f, defects of an
    exactly integrable trajectory."""
    import scipy.sparse as sp
    eng = NLPEngine(_vanderpol(), device=0)
    xl, xu, _, _ = eng.get_bounds_info()
    x = problems.seeded_iterate(eng.get_starting_point(), xl, xu, 4, "uniform")
    i, j = eng.eval_jac_g_structure()
    J = sp.coo_matrix((eng.eval_jac_g(x), (i, j)), shape=(eng.m, eng.n)).tocsr()
    rng = np.random.RandomState(0)
    for _ in range(3):
        dx = rng.uniform(-1, 1, eng.n)
        e = 1e-6
        fd = (eng.eval_g(x + e * dx) - eng.eval_g(x - e * dx)) / (2 * e)
        assert np.max(np.abs(J @ dx - fd)) <= 2e-4 * max(1.0, np.max(np.abs(fd)))
        fdf = (eng.eval_f(x + e * dx) - eng.eval_f(x - e * dx)) / (2 * e)
        assert abs(eng.eval_grad_f(x) @ dx - fdf) <= 2e-4 * max(1.0, abs(fdf))
    # x1 = x2 = u = 0 is an equilibrium: every defect vanishes, the cost is zero
    z = np.zeros(eng.n)
    z[-2:] = x[-2:]           # t0, tf
    assert np.max(np.abs(eng.eval_g(z)[:eng.m - 1])) == 0.0 and eng.eval_f(z) == 0.0
    eng.close()

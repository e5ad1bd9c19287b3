"""hessian-approximation = limited-memory — what lpopc configures by default (Core/LpNLPWrapper.hpp:71, handed to Ipopt at
Core/LpNLPSolver.cpp:27-33) — on the device solver (rpm_ipm_*, csrc/rpm_ipm_lbfgs.hip) and in its restatement
(oracle/ipm_oracle.py): Ipopt's limited-memory BFGS (history 6, sigma = s'y / s's, skipping rule, compact representation) in
the place of the exact Hessian, the low-rank part of the KKT matrix handled by the Sherman-Morrison-Woodbury formula on top
of the factorisation of the diagonal-Hessian matrix.  Ipopt is absent from the reference tree: parity is device against
restatement (same iteration counts, same objective) and both against closed-form optima."""
import numpy as np
import pytest

from lpopc_amd import problems
from oracle import ipm_oracle
from oracle import oracle as orc

CASES = [("param_sled", lambda: problems.param_sled(2, 12), 2.5, 1e-6),
         ("brachistochrone", lambda: problems.brachistochrone(2, 10), None, None),
         ("bryson_denham_10x4", lambda: problems.bryson_denham(10, 4), 4.0, 2e-2),
         ("param_oscillator", lambda: problems.param_oscillator(), None, None)]


@pytest.mark.parametrize("name,make,optimum,tol", CASES, ids=[c[0] for c in CASES])
def test_restatement_with_the_limited_memory_hessian(name, make, optimum, tol):
    o = orc.Oracle(make())
    r = ipm_oracle.solve(o, o.starting_point(), hessian_approximation="limited-memory")
    assert r["status"] == 0 and r["lm_updates"] >= r["iterations"] - 3
    if optimum is not None:
        assert abs(r["obj"] - optimum) < tol, r["obj"]
    # the same optimum as with lpopc's exact (finite-difference) Hessian (defined without static parameters; the coarse
    # Bryson-Denham mesh has several KKT points, the two Hessians end at different ones)
    if name == "brachistochrone":
        from lpopc_amd.problem import Options
        ex = Options()
        ex.SetStringValue("hessian-approximation", "exact")
        re = ipm_oracle.solve(orc.Oracle(make(), ex), o.starting_point())
        assert re["status"] == 0 and abs(re["obj"] - r["obj"]) < 1e-5 * max(1.0, abs(re["obj"]))


@pytest.mark.gpu
@pytest.mark.parametrize("name,make,optimum,tol", CASES, ids=[c[0] for c in CASES])
def test_device_limited_memory_against_the_restatement(built, name, make, optimum, tol):
    from lpopc_amd.engine import BatchedIPM, NLPEngine
    prob = make()
    eng = NLPEngine(prob, device=0)                 # default options: hessian-approximation = limited-memory
    ipm = BatchedIPM(eng, trace=256)
    o = orc.Oracle(prob)
    x0 = o.starting_point()
    r = ipm.solve(x0[None, :])
    ro = ipm_oracle.solve(o, x0, hessian_approximation="limited-memory")
    assert int(r["status"][0]) == 0 and ro["status"] == 0
    assert abs(int(r["iterations"][0]) - ro["iterations"]) <= 2, (int(r["iterations"][0]), ro["iterations"])
    assert abs(float(r["obj"][0]) - ro["obj"]) <= 1e-8 * max(1.0, abs(ro["obj"]))
    assert np.max(np.abs(r["x"][0] - ro["x"])) <= 1e-5 * max(1.0, np.max(np.abs(ro["x"])))
    # step by step: the first accepted steps (no pair, one pair, two pairs, ...) agree closely
    tr = ipm.trace(0, 256)
    for k in range(min(5, len(tr), len(ro["trace"]))):
        assert abs(tr[k][0] - ro["trace"][k]["f"]) <= 1e-6 * max(1.0, abs(ro["trace"][k]["f"])), k
        assert abs(tr[k][3] - ro["trace"][k]["alpha"]) <= 1e-5, k
    if optimum is not None:
        assert abs(float(r["obj"][0]) - optimum) < tol
    st = ipm.stats()
    assert st["factorizations"] <= st["iterations"] + 2      # a positive definite approximation: no inertia corrections
    ipm.close()
    eng.close()


@pytest.mark.gpu
def test_device_limited_memory_batch_and_delta_iii(built):
    """A batch of perturbed starts (every instance its own memory), and Delta-III from lpopc's default guess with lpopc's default
    Hessian option: the published optimum 7529.71 kg."""
    from lpopc_amd.engine import BatchedIPM, NLPEngine
    prob, B = problems.param_oscillator(), 7
    eng = NLPEngine(prob, n_instances=B, device=0)
    ipm = BatchedIPM(eng)
    x0 = eng.get_starting_point()[:eng.n]
    starts = x0[None, :] * (1 + 1e-2 * np.random.RandomState(1).uniform(-1, 1, size=(B, x0.size)))
    starts[0] = x0
    r = ipm.solve(starts)
    assert (r["status"] <= 1).all() and np.ptp(r["obj"]) < 1e-6
    one = NLPEngine(prob, device=0)
    s1 = BatchedIPM(one)
    r1 = s1.solve(starts[3:4])
    assert int(r1["iterations"][0]) == int(r["iterations"][3]) and float(r1["obj"][0]) == float(r["obj"][3])   # instances are independent
    for h in (s1, ipm):
        h.close()
    for e in (one, eng):
        e.close()
    eng = NLPEngine(problems.launch(8, 8), device=0)
    ipm = BatchedIPM(eng, max_iter=3000)
    r = ipm.solve(eng.get_starting_point()[None, :])
    assert int(r["status"][0]) <= 1 and abs(-float(r["obj"][0]) * 301454.0 - 7529.71) < 0.1
    ipm.close()
    eng.close()


@pytest.mark.gpu
def test_lpopc_default_options_reach_the_device_solver(built):
    """LpopcApplication with lpopc's own default options (hessian-approximation = limited-memory, Core/LpNLPWrapper.hpp:71): every
    NLP of the mesh-refinement loop is solved on the device — including a problem with a static parameter."""
    from lpopc_amd.application import DeviceIPMSolver, LpopcApplication
    app = LpopcApplication(0)
    app.SetOptimalControlProblem(problems.param_sled(2, 12))
    app.Options().SetIntegerValue("max-grid-num", 3)
    seen = []
    solver = DeviceIPMSolver(app.Options().GetNumericValue("Ipopt-tol"))
    orig = solver.SolveNlp

    def spy(nlp):
        ok = orig(nlp)
        seen.append((ok, solver.attempts[-1]))
        return ok
    solver.SolveNlp = spy
    try:
        app.SolveOptimalProblem(nlp_solver=solver)
    except Exception as ex:                        # "reach the max number of refine grid" is the reference's own ending for a bang-bang control
        assert "max number" in str(ex)
    assert seen and all(ok for ok, _ in seen)
    assert abs(app.objective - 2.5) < 1e-3
    # and without an explicit solver the application picks the device solver for either Hessian option
    app2 = LpopcApplication(0)
    app2.SetOptimalControlProblem(problems.brachistochrone(2, 10))
    app2.Options().SetIntegerValue("max-grid-num", 2)
    try:
        app2.SolveOptimalProblem()
    except Exception as ex:
        assert "max number" in str(ex)
    assert isinstance(app2.last_solver, DeviceIPMSolver) and abs(app2.objective - 0.82448) < 1e-3

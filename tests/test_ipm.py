"""Row f-2: the NLP solve, batched and device-resident (rpm_ipm_*), against its CPU restatement (oracle/ipm_oracle.py).

The reference's solver is Ipopt 3.12.3 (Core/LpNLPSolver.cpp:13-53), a third-party dependency that is not in the
reference tree; both sides here restate its published algorithm, so parity is device-vs-restatement on the same
inputs (same iteration counts, same optimum to solver tolerance) and the restatement itself is pinned by optima known
in closed form and by scipy's trust-constr — not by traces of the reference, which holds none.

CPU (not gpu): the restatement reaches the analytic optima; agrees with scipy.
GPU: band + border LDL^T vs numpy; device solves vs the restatement (smooth problems: identical iteration counts);
per-instance bounds (an MPC sweep over initial states); device-pointer entry; the application loop with the device
solver, whose multipliers give the analytic Bryson-Denham costates.
"""
import numpy as np
import pytest

from lpopc_amd import problems
from lpopc_amd.problem import Options
from oracle import ipm_oracle
from oracle import oracle as orc


def _exact():
    o = Options()
    o.SetStringValue("hessian-approximation", "exact")
    return o


def _cycloid_time(xf, yf, g):
    from scipy.optimize import brentq
    th = brentq(lambda t: (t - np.sin(t)) / (1 - np.cos(t)) - xf / yf, 1e-6, 2 * np.pi - 1e-6)
    return th * np.sqrt(yf / (1 - np.cos(th)) / g)


# ----------------------------------------------------------------------------------------------- CPU: the restatement
def test_restatement_reaches_bryson_denham_optimum():
    o = orc.Oracle(problems.bryson_denham(2, 8), _exact())
    r = ipm_oracle.solve(o, o.starting_point())
    assert r["status"] == 0 and r["kkt_error"] <= 1e-8
    assert abs(r["obj"] - 4.0) < 1e-6                      # J* = 4 / (9 l), l = 1/9
    M = 17
    assert r["x"][:M].max() <= 1.0 / 9.0 + 1.001e-8         # the state rides its bound (moved out by bound_relax_factor 1e-8, as in Ipopt)


def test_restatement_reaches_brachistochrone_optimum():
    prob = problems.brachistochrone(2, 10)
    o = orc.Oracle(prob, _exact())
    r = ipm_oracle.solve(o, o.starting_point())
    ev = prob.GetPhase(0).GeteventMin()
    T = _cycloid_time(ev[3], ev[4], prob.GetOpimalProblemFuns().consts[0])
    assert r["status"] == 0 and abs(r["obj"] - T) < 1e-5 * T


def test_restatement_reaches_hypersensitive_turnpike_cost():
    from scipy.integrate import quad
    V = quad(lambda x: -x ** 3 + np.sqrt(x ** 6 + x ** 2), 0, 1.5)[0]
    W = quad(lambda x: x ** 3 + np.sqrt(x ** 6 + x ** 2), 0, 1.0)[0]
    o = orc.Oracle(problems.hypersensitive(np.linspace(-1, 1, 7).tolist(), [10] * 6, tf=30.0), _exact())
    r = ipm_oracle.solve(o, o.starting_point())
    assert r["status"] == 0 and abs(r["obj"] - (V + W)) < 1e-3 * (V + W)


def test_restatement_agrees_with_scipy_on_the_quadrotor():
    from lpopc_amd.application import ScipyNLPSolver
    from test_known_answers import _OracleNLP
    o = orc.Oracle(problems.quadrotor(2, 4), _exact())
    r = ipm_oracle.solve(o, o.starting_point())
    nlp = _OracleNLP(o)
    assert ScipyNLPSolver(1e-8, maxiter=400).SolveNlp(nlp)
    assert r["status"] == 0 and abs(r["obj"] - nlp.sol[2]) < 1e-6 * abs(nlp.sol[2])
    # KKT conditions of the restatement's answer, checked from scratch: stationarity in the free variables
    xl, xu, gl, gu = o.bounds()
    ji, jj = o.jac_structure()
    lag = o.eval_grad_f(r["x"])
    np.add.at(lag, jj, o.eval_jac_g(r["x"]) * r["lambda"][ji])
    inner = (r["x"] > xl + 1e-6) & (r["x"] < xu - 1e-6)
    assert np.abs(lag[inner]).max() < 1e-6


# ----------------------------------------------------------------------------------------------- GPU
def _random_kkt(ipm, n, B, seed):
    """Random symmetric quasi-definite matrices inside the solver's band + border envelope -> (storage, dense, signs)."""
    info = ipm.info()
    nt, nbo, b = info["kkt_order"], info["band_order"], info["half_bandwidth"]
    cs = info["storage_doubles"] // nt
    pos = ipm.permutation()
    sign = np.ones(nt)
    sign[pos[n + info["n_slacks"]:]] = -1.0
    rng = np.random.RandomState(seed)
    ii, jj = np.tril_indices(nt, -1)
    inside = ((ii < nbo) & (ii - jj <= b)) | (ii >= nbo)
    dense, store = np.zeros((B, nt, nt)), np.zeros((B, nt * cs))
    for bi in range(B):
        keep = inside & ((sign[ii] != sign[jj]) | (rng.rand(ii.size) < 0.3)) & (rng.rand(ii.size) < 0.5)
        A = np.zeros((nt, nt))
        A[ii[keep], jj[keep]] = rng.uniform(-1, 1, size=keep.sum())
        A = A + A.T
        same = sign[:, None] == sign[None, :]
        A[np.arange(nt), np.arange(nt)] = sign * ((np.abs(A) * same).sum(axis=1) + rng.uniform(0.5, 2.0, size=nt))
        dense[bi] = A
        li, lj = np.tril_indices(nt)
        ok = ((li < nbo) & (li - lj <= b)) | (li >= nbo)
        li, lj = li[ok], lj[ok]
        slot = np.where(li < nbo, li - lj, b + 1 + li - nbo)
        store[bi, lj * cs + slot] = A[li, lj]
    return store, dense, sign


LAYOUTS = [("brachistochrone", lambda: problems.brachistochrone(2, 6), 3), ("quadrotor", lambda: problems.quadrotor(3, 4), 2),
           ("launch", lambda: problems.launch(2, 5), 2), ("hypersensitive_hp", lambda: problems.hypersensitive([-1, -0.5, 0.4, 1], [4, 9, 3], tf=50.0), 2)]


@pytest.mark.gpu
@pytest.mark.parametrize("name,make,B", LAYOUTS, ids=[c[0] for c in LAYOUTS])
def test_band_border_ldlt_against_numpy(built, name, make, B):
    from lpopc_amd.engine import BatchedIPM, NLPEngine
    eng = NLPEngine(make(), _exact(), n_instances=B, device=0)
    eng.set_option("ipm_nested", 0)          # this test fills the band + border storage itself
    ipm = BatchedIPM(eng)
    store, dense, sign = _random_kkt(ipm, eng.n, B, 7)
    rhs = np.random.RandomState(3).uniform(-1, 1, size=(B, sign.size))
    sol, npos, nneg = ipm.debug_solve(store, rhs)
    for bi in range(B):
        ref = np.linalg.solve(dense[bi], rhs[bi])
        assert np.max(np.abs(sol[bi] - ref)) <= 1e-11 * np.max(np.abs(ref))      # tolerance: f64 LDL^T of a well-conditioned matrix
        assert npos[bi] == (sign > 0).sum() and nneg[bi] == (sign < 0).sum()       # Sylvester: signs of D = inertia
    ipm.close()
    eng.close()


def _random_kkt_dense(ipm, n, B, seed):
    """Random symmetric quasi-definite matrices inside the envelope of whatever layout the solver uses (asked entry by
    entry through rpm_ipm_debug_slot) -> (dense (B, Nt, Nt) in unknown order, signs)."""
    info = ipm.info()
    nt = info["kkt_order"]
    sign = np.ones(nt)
    sign[n + info["n_slacks"]:] = -1.0
    ii, jj = np.tril_indices(nt, -1)
    inside = np.array([ipm.slot(int(a), int(c)) >= 0 for a, c in zip(ii, jj)])
    assert all(ipm.slot(a, a) >= 0 for a in range(nt))
    rng = np.random.RandomState(seed)
    dense = np.zeros((B, nt, nt))
    for bi in range(B):
        keep = inside & ((sign[ii] != sign[jj]) | (rng.rand(ii.size) < 0.3)) & (rng.rand(ii.size) < 0.5)
        A = np.zeros((nt, nt))
        A[ii[keep], jj[keep]] = rng.uniform(-1, 1, size=keep.sum())
        A = A + A.T
        same = sign[:, None] == sign[None, :]
        A[np.arange(nt), np.arange(nt)] = sign * ((np.abs(A) * same).sum(axis=1) + rng.uniform(0.5, 2.0, size=nt))
        dense[bi] = A
    return dense, sign, int(inside.sum())


ND_LAYOUTS = LAYOUTS + [("launch_ragged", lambda: problems.launch(), 1), ("launch_unequal", lambda: problems.launch(), 1), ("quadrotor_8x8", lambda: problems.quadrotor(8, 8), 2),
                        ("quadrotor_24x4", lambda: problems.quadrotor(24, 4), 2), ("launch_16x4", lambda: problems.launch(16, 4), 1)]


@pytest.mark.gpu
@pytest.mark.parametrize("nested", [0, 1, 2], ids=["band", "nested", "nested_3_levels"])
@pytest.mark.parametrize("name,make,B", ND_LAYOUTS, ids=[c[0] for c in ND_LAYOUTS])
def test_kkt_factorisation_layouts_against_numpy(built, name, make, B, nested):
    """Band + border LDL^T and the nested dissection over the mesh intervals (every interval eliminated by its own
    workgroup, Schur complements summed into the block-tridiagonal separator system) against numpy.linalg.solve on random
    quasi-definite matrices that fill each layout's envelope; the signs of D give the inertia in both."""
    from lpopc_amd.engine import BatchedIPM, NLPEngine
    prob = make()
    if name == "launch_ragged":
        # (equal node totals per phase: with unequal ones lpopc's link Hessian indexes the right phase's initial states with the
        # LEFT phase's node count, LpHessian.cpp:1150, kept bug-for-bug, and those entries land in the middle of a phase)
        meshes = [([-1, -0.6, 0.1, 1], [5, 8, 2]), ([-1, 0.5, 1], [7, 8]), ([-1, 1], [15]), ([-1, -0.9, -0.5, 0.0, 1], [3, 4, 3, 5])]
        for i, (mesh, nodes) in enumerate(meshes):
            problems.set_mesh(prob.GetPhase(i), mesh, nodes)
    if name == "launch_unequal":
        # unequal node totals (what hp-Liu refinement produces): the mis-indexed link-Hessian entries couple states of different
        # nodes of the right phase; their endpoints are promoted to the border (promoted_to_border, rpm_ipm.cpp)
        meshes = [([-1, -0.6, 0.1, 1], [5, 8, 4]), ([-1, 0.5, 1], [7, 6]), ([-1, 0.2, 1], [6, 5]), ([-1, -0.9, -0.5, 0.0, 1], [3, 4, 6, 5])]
        for i, (mesh, nodes) in enumerate(meshes):
            problems.set_mesh(prob.GetPhase(i), mesh, nodes)
    eng = NLPEngine(prob, _exact(), n_instances=B, device=0)
    eng.set_option("ipm_nested", min(nested, 1))       # (launch_unequal: the promoted unknowns bring the border to 107 rows, 121 in an interval block)
    if nested == 2:                  # the separator system cut once more, into groups of 48 of its positions (at least 3 bandwidths)
        eng.set_option("ipm_nested_group", 48)
    ipm = BatchedIPM(eng)
    if nested == 2 and name in ("quadrotor_24x4", "launch_16x4"):
        n_intervals = sum(len(prob.GetPhase(i).GetNodesPerInterval()) for i in range(prob.GetPhaseNum()))
        assert ipm.subproblems().shape[0] >= n_intervals + 3               # intervals, >= 2 groups, last level
    dense, sign, filled = _random_kkt_dense(ipm, eng.n, B, 11)
    assert filled > 4 * sign.size
    rhs = np.random.RandomState(5).uniform(-1, 1, size=(B, sign.size))
    sol, npos, nneg = ipm.debug_solve_dense(dense, rhs)
    for bi in range(B):
        ref = np.linalg.solve(dense[bi], rhs[bi])
        assert np.max(np.abs(sol[bi] - ref)) <= 1e-11 * np.max(np.abs(ref))
        assert npos[bi] == (sign > 0).sum() and nneg[bi] == (sign < 0).sum()
    ipm.close()
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,make,B", [("quadrotor_8x8", lambda: problems.quadrotor(8, 8), 3), ("quadrotor_24x4", lambda: problems.quadrotor(24, 4), 2),
                                         ("launch_16x4", lambda: problems.launch(16, 4), 1), ("launch_2x16", lambda: problems.launch(2, 16), 1),
                                         ("launch_1x18", lambda: problems.launch(1, 18), 1)],
                         ids=["quadrotor_8x8", "quadrotor_24x4", "launch_16x4", "launch_2x16", "launch_1x18"])
def test_register_resident_level_1_equals_the_left_looking_kernel(built, name, make, B):
    """kkt_factor_dense_kernel (interval blocks factored out of registers: the trailing 17 block rows resident, up to 4 block columns
    before them — the metric problem's intervals have 2, launch_1x18's 4 — taken through the storage; option level1_dense) against
    kkt_factor_kernel on the same random quasi-definite matrices: both solve to 1e-11 of numpy, report the same inertia, and agree
    with each other bit for bit (the same products in the same order).  A layout without nested dissection refuses the option."""
    from lpopc_amd.engine import BatchedIPM, NLPEngine, RpmError
    eng = NLPEngine(make(), _exact(), n_instances=B, device=0)
    eng.set_option("ipm_nested", 1)
    ipm = BatchedIPM(eng)
    dense, sign, filled = _random_kkt_dense(ipm, eng.n, B, 23)
    rhs = np.random.RandomState(9).uniform(-1, 1, size=(B, sign.size))
    sols = []
    for on in (1, 0):
        ipm.set_option("level1_dense", on)          # 1 is the default here; it must be accepted
        sol, npos, nneg = ipm.debug_solve_dense(dense, rhs)
        for bi in range(B):
            ref = np.linalg.solve(dense[bi], rhs[bi])
            assert np.max(np.abs(sol[bi] - ref)) <= 1e-11 * np.max(np.abs(ref))
            assert npos[bi] == (sign > 0).sum() and nneg[bi] == (sign < 0).sum()
        sols.append(sol)
    assert np.array_equal(sols[0], sols[1])     # bit for bit: Delta-III's path is sensitive to the last bit of L D
    ipm.close()
    eng.close()
    band = NLPEngine(make(), _exact(), n_instances=1, device=0)
    band.set_option("ipm_nested", 0)
    ipm = BatchedIPM(band)
    with pytest.raises(RpmError):
        ipm.set_option("level1_dense", 1)
    ipm.set_option("level1_dense", 0)
    ipm.close()
    band.close()


@pytest.mark.gpu
@pytest.mark.parametrize("group", [0, 48], ids=["two_levels", "three_levels"])
@pytest.mark.parametrize("name,make,B", [("quadrotor_24x4", lambda: problems.quadrotor(24, 4), 2), ("launch_16x4", lambda: problems.launch(16, 4), 1),
                                         ("launch_2x16", lambda: problems.launch(2, 16), 1)], ids=["quadrotor_24x4", "launch_16x4", "launch_2x16"])
def test_register_resident_upper_levels(built, name, make, B, group):
    """Option upper_dense (1, the default where the sub-problems fit): the last level (and wide-band groups of separators) on
    kkt_factor_dense_kernel, the corner's block columns included (their panels by substitution: agrees with kkt_factor_kernel to
    rounding); 2: the band part there, the corner by kkt_factor_kernel's unblocked elimination (partial = 2) — bit for bit what
    kkt_factor_kernel alone gives.  All solve to 1e-11 of numpy with the same inertia."""
    from lpopc_amd.engine import BatchedIPM, NLPEngine
    eng = NLPEngine(make(), _exact(), n_instances=B, device=0)
    eng.set_option("ipm_nested", 1)
    if group:
        eng.set_option("ipm_nested_group", group)
    ipm = BatchedIPM(eng)
    dense, sign, filled = _random_kkt_dense(ipm, eng.n, B, 29)
    rhs = np.random.RandomState(13).uniform(-1, 1, size=(B, sign.size))
    sols = []
    for on in (2, 0, 1):
        ipm.set_option("upper_dense", on)
        sol, npos, nneg = ipm.debug_solve_dense(dense, rhs)
        for bi in range(B):
            ref = np.linalg.solve(dense[bi], rhs[bi])
            assert np.max(np.abs(sol[bi] - ref)) <= 1e-11 * np.max(np.abs(ref))
            assert npos[bi] == (sign > 0).sum() and nneg[bi] == (sign < 0).sum()
        sols.append(sol)
    assert np.array_equal(sols[0], sols[1])
    assert np.max(np.abs(sols[2] - sols[1])) <= 1e-11 * np.max(np.abs(sols[1]))
    ipm.close()
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("hessian", ["exact", "limited-memory"])
def test_level_1_assembled_in_the_factor_kernel_changes_nothing(built, hessian):
    """Option fused_fill (default where kkt_factor_dense_kernel runs): the interval blocks are built from the Jacobian, Hessian
    and diagonal terms inside the factorisation kernel and the fill leaves their storage alone.  Same matrix entries, same products:
    every instance of a sweep (regular iterations, inertia corrections, second-order corrections, per-instance bounds) takes the
    same path and ends at the same point, bit for bit, as with the fill kernel writing the whole storage."""
    from lpopc_amd.engine import BatchedIPM, NLPEngine, RpmError
    from lpopc_amd.problem import Options
    B = 5
    prob = problems.quadrotor(8, 8)
    o = Options()
    o.SetStringValue("hessian-approximation", hessian)
    out = []
    for fused in (1, 0):
        eng = NLPEngine(prob, o, n_instances=B, device=0)
        eng.set_option("ipm_nested", 1)
        ipm = BatchedIPM(eng, max_iter=300, trace=300)
        ipm.set_option("fused_fill", fused)        # 1 is the default here; it must be accepted
        xl, xu, _, _ = eng.get_bounds_info()
        rng = np.random.RandomState(11)
        N1 = 8 * 8 + 1
        for bi in range(B):
            l, u = xl.copy(), xu.copy()
            l[[i * N1 for i in range(12)]] = u[[i * N1 for i in range(12)]] = rng.uniform(-0.4, 0.4, 12)
            ipm.set_bounds(bi, l, u)
        x0 = np.tile(eng.get_starting_point(), (B, 1))
        r = ipm.solve(x0)
        out.append((r, [ipm.trace(bi).copy() for bi in range(B)], ipm.stats()))
        ipm.close()
        eng.close()
    (a, ta, sa), (b, tb, sb) = out
    if hessian == "exact":                         # (the limited-memory runs of this sweep end at the acceptable level or in a failed
        assert (a["status"] == 0).all()            #  line search: paths long enough to show any difference all the same)
    assert np.array_equal(a["status"], b["status"]) and np.array_equal(a["iterations"], b["iterations"])
    assert np.array_equal(a["x"], b["x"]) and np.array_equal(a["obj"], b["obj"])
    assert all(np.array_equal(p, q) for p, q in zip(ta, tb))
    assert sa["factorizations"] == sb["factorizations"]
    band = NLPEngine(problems.quadrotor(2, 4), o, n_instances=1, device=0)
    band.set_option("ipm_nested", 0)
    ipm = BatchedIPM(band)
    with pytest.raises(RpmError):
        ipm.set_option("fused_fill", 1)            # no level 1, no register-resident kernel
    ipm.close()
    band.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,make,B,pert", [("bryson_denham", lambda: problems.bryson_denham(2, 8), 2, 0.0),
                                             ("hypersensitive", lambda: problems.hypersensitive(np.linspace(-1, 1, 7).tolist(), [10] * 6, tf=30.0), 2, 0.0),
                                             ("quadrotor_3x6", lambda: problems.quadrotor(3, 6, pref=(0.4, 0.8, -0.6)), 3, 2e-2),
                                             ("quadrotor_8x8", lambda: problems.quadrotor(8, 8), 4, 1e-2)],
                         ids=["bryson_denham", "hypersensitive", "quadrotor_3x6", "quadrotor_8x8"])
def test_device_solve_nested_dissection_equals_band(built, name, make, B, pert):
    """The whole interior-point solve with the nested-dissection factorisation: same verdicts, iteration counts and optima
    as with the band + border factorisation (the two eliminate in different orders, so iterates agree to rounding only)."""
    from lpopc_amd.engine import BatchedIPM, NLPEngine
    prob = make()
    out = []
    for nested in (0, 1):
        eng = NLPEngine(prob, _exact(), n_instances=B, device=0)
        eng.set_option("ipm_nested", nested)
        x0 = np.tile(eng.get_starting_point(), (B, 1))
        if pert:
            x0 = x0 * (1 + pert * np.random.RandomState(1).uniform(-1, 1, size=x0.shape))
        ipm = BatchedIPM(eng, max_iter=400)
        out.append(ipm.solve(x0))
        ipm.close()
        eng.close()
    a, b = out
    assert (a["status"] == 0).all() and (b["status"] == 0).all()
    assert np.max(np.abs(a["iterations"].astype(int) - b["iterations"].astype(int))) <= 1
    assert np.max(np.abs(a["obj"] - b["obj"]) / np.maximum(1.0, np.abs(a["obj"]))) <= 1e-8
    # both stop at E_0 <= 1e-8, a step apart at most: the flat directions of the tracking cost move x by ~1e-5 between them
    assert np.max(np.abs(a["x"] - b["x"])) <= 1e-5 * max(1.0, np.max(np.abs(a["x"])))


SOLVES = [
    # name, problem, instances, relative start perturbation, identical iteration counts expected
    ("bryson_denham", lambda: problems.bryson_denham(2, 8), 2, 0.0, True),
    ("brachistochrone", lambda: problems.brachistochrone(2, 10), 2, 0.0, True),
    ("hypersensitive", lambda: problems.hypersensitive(np.linspace(-1, 1, 7).tolist(), [10] * 6, tf=30.0), 2, 0.0, True),
    ("quadrotor", lambda: problems.quadrotor(2, 4), 5, 1e-2, True),
    ("quadrotor_3x6", lambda: problems.quadrotor(3, 6, pref=(0.4, 0.8, -0.6)), 3, 2e-2, True),
]


@pytest.mark.gpu
@pytest.mark.parametrize("name,make,B,pert,same_path", SOLVES, ids=[c[0] for c in SOLVES])
def test_device_solve_against_restatement(built, name, make, B, pert, same_path):
    from lpopc_amd.engine import BatchedIPM, NLPEngine
    prob = make()
    eng = NLPEngine(prob, _exact(), n_instances=B, device=0)
    o = orc.Oracle(prob, _exact())
    x0 = np.tile(o.starting_point(), (B, 1))
    if pert:
        x0 = x0 * (1 + pert * np.random.RandomState(1).uniform(-1, 1, size=x0.shape))
    ipm = BatchedIPM(eng, max_iter=400, trace=400)
    r = ipm.solve(x0)
    for bi in range(B):
        ref = ipm_oracle.solve(o, x0[bi], max_iter=400)
        assert r["status"][bi] == ref["status"] == 0
        assert abs(r["obj"][bi] - ref["obj"]) <= 1e-8 * max(1.0, abs(ref["obj"]))          # both stop at E_0 <= 1e-8
        assert r["kkt_error"][bi] <= 1e-8
        if same_path:
            assert abs(int(r["iterations"][bi]) - ref["iterations"]) <= 1      # a barrier update decided at rounding level may shift by one
            assert np.max(np.abs(r["x"][bi] - ref["x"])) <= 1e-6 * max(1.0, np.max(np.abs(ref["x"])))
            # (rows that only fixed variables enter — e.g. an event on a fixed end point — have no Jacobian entry among the free
            # unknowns: their multiplier is c / delta_c of a residual at rounding level, 1e5 with nothing behind it; left out)
            ji, jj = o.jac_structure()
            xl_, xu_, _, _ = o.bounds()
            live = np.zeros(o.m, dtype=bool)
            live[ji[xl_[jj] != xu_[jj]]] = True
            dl = np.abs(r["lambda"][bi] - ref["lambda"])[live]
            # (1e-3: Bryson-Denham and the brachistochrone carry multipliers of 1e5 on rows the optimum barely depends on; the two
            # paths end a rounding-level barrier parameter apart and those multipliers 3e-5 .. 3.7e-4 apart, depending on the
            # order in which the factorisation adds things up)
            assert dl.size == 0 or np.max(dl) <= 1e-3 * max(1.0, np.max(np.abs(ref["lambda"][live])))
            # step by step, while both are on the same path (a decision taken at rounding level may part them late): same
            # barrier parameter, inertia corrections and backtracking counts.  The first step uses lambda = 0, so its
            # Hessian is the objective's alone and everything agrees to rounding; from the second step on the constraint
            # Hessians enter — second differences whose CPU and GPU values differ by 1e-9 (polynomial dynamics) to 1e-3
            # (libm calls, HESS_CASES in test_gpu_parity.py) — and the two Newton paths run ~1e-4 apart to the same optimum
            tr = ipm.trace(bi)
            same = min(len(tr), len(ref["trace"]), 5)
            assert same >= min(5, ref["iterations"])
            for k in range(same):
                e = ref["trace"][k]
                # step 0: both solve the same KKT system (condition ~1e11 with delta_c = 1e-9 next to barrier terms of 1e2) in
                # different elimination orders (dense LU there, interval-wise nested dissection here): step lengths to 1e-5
                rel, ab = (1e-8, 1e-5) if k == 0 else (1e-2, 1e-2)
                # (mu comes from the adaptive rule — the iterate's average and smallest complementarity — not from a fixed sequence)
                # — sigma goes with the cube of (1 - xi) / xi, so a 1 % difference in the smallest complementarity is 3 % in mu)
                assert abs(tr[k, 2] - e["mu"]) <= (1e-6 if k < 2 else 1e-1) * e["mu"] and abs(tr[k, 5] - e["delta_w"]) <= 1e-12 * e["delta_w"], (k, tr[k], e)
                assert int(tr[k, 7]) == e["ls"], (k, tr[k], e)
                assert abs(tr[k, 0] - e["f"]) <= rel * max(1.0, abs(e["f"])), (k, tr[k], e)
                assert abs(tr[k, 1] - e["theta"]) <= rel * max(1.0, e["theta"]), (k, tr[k], e)
                assert abs(tr[k, 3] - e["alpha"]) <= ab and abs(tr[k, 4] - e["alpha_z"]) <= ab, (k, tr[k], e)
    # every instance satisfies its bounds and constraints
    xl, xu, gl, gu = o.bounds()
    for bi in range(B):
        g = o.eval_g(r["x"][bi])
        rl, ru = 1.001e-8 * np.maximum(1.0, np.abs(xl)), 1.001e-8 * np.maximum(1.0, np.abs(xu))   # bound_relax_factor, as in Ipopt
        assert (r["x"][bi] >= xl - rl).all() and (r["x"][bi] <= xu + ru).all()
        assert max((gl - g).max(), (g - gu).max()) < 1e-7
    ipm.close()
    eng.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,make", [("bryson_denham", lambda: problems.bryson_denham(2, 8)), ("brachistochrone", lambda: problems.brachistochrone(2, 10))],
                         ids=["bryson_denham", "brachistochrone"])
def test_inertia_correction_hot_start_against_restatement(built, name, make):
    """Option ic_hot_start (an experiment, not Ipopt's rule; off by default): an iteration whose predecessor needed delta_w > 0
    starts Algorithm IC at kappa_w^- delta_w_last instead of 0.  Device and restatement take the same perturbations step by step
    and reach the same optimum; nearly every iteration is then factored with delta_w > 0, against a minority without the option."""
    from lpopc_amd.engine import BatchedIPM, NLPEngine
    prob = make()
    eng = NLPEngine(prob, _exact(), n_instances=1, device=0)
    o = orc.Oracle(prob, _exact())
    x0 = o.starting_point()[None, :]
    ipm = BatchedIPM(eng, max_iter=400, trace=400, ic_hot_start=1)
    r = ipm.solve(x0)
    ref = ipm_oracle.solve(o, x0[0], max_iter=400, ic_hot_start=1)
    plain = ipm_oracle.solve(o, x0[0], max_iter=400)
    assert r["status"][0] == ref["status"] == 0
    assert abs(int(r["iterations"][0]) - ref["iterations"]) <= 1
    assert abs(r["obj"][0] - ref["obj"]) <= 1e-8 * max(1.0, abs(ref["obj"])) and abs(ref["obj"] - plain["obj"]) <= 1e-7 * max(1.0, abs(plain["obj"]))
    tr = ipm.trace(0)
    same = min(len(tr), len(ref["trace"]), 8)
    for k in range(same):
        dw = ref["trace"][k]["delta_w"]
        assert abs(tr[k, 5] - dw) <= 1e-12 * dw, (k, tr[k], ref["trace"][k])
    hot = sum(1 for e in ref["trace"] if e["delta_w"] > 0)
    assert hot > 2 * sum(1 for e in plain["trace"] if e["delta_w"] > 0) and hot >= 0.8 * len(ref["trace"])
    assert ipm.stats()["factorizations"] < 1.3 * int(r["iterations"][0]) + 3      # the failed trial at delta_w = 0 is gone
    ipm.close()
    eng.close()


SCALED = [("hypersensitive", lambda: problems.hypersensitive(np.linspace(-1, 1, 7).tolist(), [10] * 6, tf=30.0)),
          ("bryson_denham_8x6", lambda: problems.bryson_denham(8, 6)), ("brachistochrone_6x8", lambda: problems.brachistochrone(6, 8))]


@pytest.mark.parametrize("name,make", SCALED, ids=[c[0] for c in SCALED])
def test_restatement_gradient_based_scaling(name, make):
    """Ipopt's default NLP scaling (nlp_scaling_method = gradient-based), an option of this restatement: on these meshes some
    defect rows have differentiation-matrix entries above 100 at the starting point and are scaled down; the scaled solve reaches
    the optimum of the unscaled one and returns the multipliers of the caller's rows."""
    prob = make()
    o = orc.Oracle(prob, _exact())
    x0 = o.starting_point()
    plain = ipm_oracle.solve(o, x0, max_iter=400)
    scaled = ipm_oracle.solve(o, x0, max_iter=400, nlp_scaling_method="gradient-based")
    assert plain["status"] == scaled["status"] == 0
    assert abs(plain["obj"] - scaled["obj"]) <= 1e-7 * max(1.0, abs(plain["obj"]))
    ji, jj = o.jac_structure()
    rmax = np.zeros(o.m)
    np.maximum.at(rmax, ji, np.abs(o.eval_jac_g(x0)))
    assert (rmax > 100).sum() >= 10                               # the scaling is not the identity here
    xl, xu, gl, gu = o.bounds()
    g = o.eval_g(scaled["x"])
    assert max((gl - g).max(), (g - gu).max()) < 1e-6             # feasible for the caller's rows
    # stationarity with the returned multipliers, in the caller's scaling: grad f + J' lambda vanishes on the variables off their bounds
    res = o.eval_grad_f(scaled["x"])
    np.add.at(res, jj, o.eval_jac_g(scaled["x"]) * scaled["lambda"][ji])
    inner = (scaled["x"] > xl + 1e-5) & (scaled["x"] < xu - 1e-5)
    assert np.max(np.abs(res[inner])) <= 1e-5 * max(1.0, np.max(np.abs(scaled["lambda"])))


@pytest.mark.gpu
@pytest.mark.parametrize("name,make", SCALED, ids=[c[0] for c in SCALED])
def test_device_gradient_based_scaling_against_restatement(built, name, make):
    """Option nlp_scaling on the device solver: factors from the gradients at the caller's starting point, every evaluation
    scaled in place (the Jacobian's constant block once), multipliers and objective handed back unscaled — step for step with
    the restatement."""
    from lpopc_amd.engine import BatchedIPM, NLPEngine
    prob = make()
    eng = NLPEngine(prob, _exact(), n_instances=1, device=0)
    o = orc.Oracle(prob, _exact())
    x0 = o.starting_point()[None, :]
    ipm = BatchedIPM(eng, max_iter=400, trace=400, nlp_scaling=1)
    r = ipm.solve(x0)
    ref = ipm_oracle.solve(o, x0[0], max_iter=400, nlp_scaling_method="gradient-based")
    assert r["status"][0] == ref["status"] == 0
    assert abs(int(r["iterations"][0]) - ref["iterations"]) <= 1
    assert abs(r["obj"][0] - ref["obj"]) <= 1e-8 * max(1.0, abs(ref["obj"]))
    assert np.max(np.abs(r["x"][0] - ref["x"])) <= 1e-6 * max(1.0, np.max(np.abs(ref["x"])))
    tr = ipm.trace(0)
    for k in range(min(len(tr), len(ref["trace"]), 4)):
        e = ref["trace"][k]
        assert abs(tr[k, 0] - e["f"]) <= 1e-2 * max(1.0, abs(e["f"])) and abs(tr[k, 1] - e["theta"]) <= 1e-2 * max(1.0, e["theta"]), (k, tr[k], e)
        assert abs(tr[k, 5] - e["delta_w"]) <= 1e-12 * e["delta_w"] and int(tr[k, 7]) == e["ls"], (k, tr[k], e)
    # the multipliers come back for the caller's rows
    ji, jj = o.jac_structure()
    xl_, xu_, _, _ = o.bounds()
    live = np.zeros(o.m, dtype=bool)
    live[ji[xl_[jj] != xu_[jj]]] = True
    dl = np.abs(r["lambda"][0] - ref["lambda"])[live]
    assert dl.size == 0 or np.max(dl) <= 3e-4 * max(1.0, np.max(np.abs(ref["lambda"][live])))
    ipm.close()
    eng.close()
    # a batch: every instance has its own factors (its own starting point), and an instance that ends early keeps its values
    B = 3
    eng = NLPEngine(prob, _exact(), n_instances=B, device=0)
    xs = np.tile(o.starting_point(), (B, 1)) * (1 + 2e-2 * np.random.RandomState(3).uniform(-1, 1, size=(B, o.n)))
    ipm = BatchedIPM(eng, max_iter=400, nlp_scaling=1)
    rb = ipm.solve(xs)
    for bi in range(B):
        refb = ipm_oracle.solve(o, xs[bi], max_iter=400, nlp_scaling_method="gradient-based")
        assert rb["status"][bi] == refb["status"] == 0
        assert abs(int(rb["iterations"][bi]) - refb["iterations"]) <= 1
        assert abs(rb["obj"][bi] - refb["obj"]) <= 1e-8 * max(1.0, abs(refb["obj"]))
    ipm.close()
    eng.close()


@pytest.mark.gpu
def test_sweep_with_per_instance_bounds(built):
    """An MPC sweep: the same transcription from different initial states (per-instance variable bounds)."""
    from lpopc_amd.engine import BatchedIPM, NLPEngine
    B = 6
    prob = problems.quadrotor(2, 4)
    eng = NLPEngine(prob, _exact(), n_instances=B, device=0)
    o = orc.Oracle(prob, _exact())
    xl, xu, _, _ = o.bounds()
    ipm = BatchedIPM(eng)
    rng = np.random.RandomState(5)
    fixed = np.nonzero(xl == xu)[0]
    N1 = 2 * 4 + 1
    x0_idx = [i for i in fixed if i % N1 == 0 and i < 12 * N1]          # X(0, state) of the 12 states
    bounds = []
    for bi in range(B):
        l, u = xl.copy(), xu.copy()
        l[x0_idx] = u[x0_idx] = rng.uniform(-0.3, 0.3, size=len(x0_idx))
        ipm.set_bounds(bi, l, u)
        bounds.append((l, u))
    x0 = np.tile(o.starting_point(), (B, 1))
    r = ipm.solve(x0)
    for bi in range(B):
        ref = ipm_oracle.solve(o, x0[bi], x_l=bounds[bi][0], x_u=bounds[bi][1])
        assert r["status"][bi] == ref["status"] == 0
        assert abs(int(r["iterations"][bi]) - ref["iterations"]) <= 1      # a barrier update decided at rounding level may shift by one
        assert abs(r["obj"][bi] - ref["obj"]) <= 1e-8 * max(1.0, abs(ref["obj"]))
        assert np.array_equal(r["x"][bi][x0_idx], bounds[bi][0][x0_idx])
    assert len(set(np.round(r["obj"], 6))) == B                          # genuinely different problems
    # a bound pattern that turns a free variable into a fixed one is refused (the KKT layout is shared)
    bad_l, bad_u = xl.copy(), xu.copy()
    free = np.nonzero(xl != xu)[0][0]
    bad_l[free] = bad_u[free] = 0.0
    with pytest.raises(Exception):
        ipm.set_bounds(0, bad_l, bad_u)
    ipm.close()
    eng.close()


@pytest.mark.gpu
def test_device_pointer_entry_and_limits(built):
    import torch
    from lpopc_amd.engine import BatchedIPM, NLPEngine, RpmError
    prob = problems.quadrotor(2, 4)
    B = 3
    eng = NLPEngine(prob, _exact(), n_instances=B, device=0)
    ipm = BatchedIPM(eng)
    x0 = np.tile(eng.get_starting_point()[:eng.n], (B, 1))
    host = ipm.solve(x0)
    d_x = torch.from_numpy(x0.copy()).cuda()
    d_l = torch.zeros((B, eng.m), dtype=torch.float64, device="cuda")
    dev = ipm.solve_dev(d_x, d_l)
    assert np.array_equal(d_x.cpu().numpy(), host["x"]) and np.array_equal(d_l.cpu().numpy(), host["lambda"])
    assert np.array_equal(dev["iterations"], host["iterations"])
    ipm.set_option("max_iter", 3)                                        # iteration limit -> status 2, no exception
    lim = ipm.solve(x0)
    assert (lim["status"] == 2).all() and (lim["iterations"] == 3).all()
    with pytest.raises(RpmError):
        ipm.set_option("no-such-option", 1.0)
    ipm.close()
    eng.close()
    # an engine with lpopc's default hessian-approximation = limited-memory gets Ipopt's limited-memory BFGS in the place of
    # eval_h (round 3, tests/test_ipm_limited_memory.py); an interval-sharded engine is refused, loudly
    lm = NLPEngine(prob, device=0)
    s_lm = BatchedIPM(lm)
    assert s_lm.info()["half_bandwidth"] > 0
    s_lm.close()
    lm.close()
    sh = NLPEngine(prob, shard_mode=1, shard_rank=0, shard_world=2, device=0)
    with pytest.raises(RpmError):
        BatchedIPM(sh)
    sh.close()


@pytest.mark.gpu
def test_application_loop_with_the_device_solver(built, tmp_path):
    """hessian-approximation=exact: LpopcApplication solves every mesh on the device (rpm_ipm_*), extracts, estimates,
    refines.  Bryson-Denham: J* = 4/(9 l); the problem is autonomous, so the Hamiltonian built from the solver's
    multipliers (Nlp2OpControl: costates = lambda / w) is constant along the trajectory."""
    from lpopc_amd.application import DeviceIPMSolver, LpopcApplication, console_not_print
    from lpopc_amd.problem import LpopcException
    app = LpopcApplication(console_not_print)
    app.SetOptimalControlProblem(problems.bryson_denham(2, 8))
    app.Options().SetStringValue("hessian-approximation", "exact")
    app.Options().SetNumericValue("Ipopt-tol", 1e-8)
    app.Options().SetIntegerValue("max-grid-num", 3)
    try:
        app.SolveOptimalProblem(device=0, result_dir=str(tmp_path))
    except LpopcException as e:                      # like the reference: running out of grids is an error exit
        assert "grid" in str(e).lower()
    assert abs(app.objective - 4.0) < 1e-5
    H = np.asarray(app.result[0]["hamiltonian"])
    assert np.ptp(H[1:-1]) < 2e-2 * max(1.0, np.abs(H).max())          # constant Hamiltonian (mesh-level accuracy)
    assert isinstance(DeviceIPMSolver(1e-8), DeviceIPMSolver)


@pytest.mark.gpu
def test_climb_at_the_reference_tolerance_and_delta3_verdict(built):
    """Minimum-time climb (table look-ups, finite-difference derivatives): the noise floor of the dual infeasibility is
    ~1e-7, so it converges at the reference's default Ipopt-tol = 1e-6 (Core/LpNLPWrapper.hpp:73), on the device and in
    the restatement, to the same optimum.  Delta-III on a small mesh with an iteration limit it may or may not reach
    (the path from lpopc's default guess is chaotic): the device solver must come back with a verdict — the published
    optimum when it says converged, a finite point and an honest status otherwise; never a hang, never garbage as converged."""
    from lpopc_amd.engine import BatchedIPM, NLPEngine
    prob = problems.min_time_climb(2, 6)
    eng = NLPEngine(prob, _exact(), device=0)
    o = orc.Oracle(prob, _exact())
    ipm = BatchedIPM(eng, tol=1e-6, max_iter=300)
    r = ipm.solve(eng.get_starting_point()[None, :])
    ref = ipm_oracle.solve(o, o.starting_point(), tol=1e-6, max_iter=300)
    assert r["status"][0] == ref["status"] == 0
    assert abs(r["obj"][0] - ref["obj"]) <= 1e-6 * abs(ref["obj"])
    ipm.close()
    eng.close()
    eng = NLPEngine(problems.launch(2, 5), _exact(), device=0)
    ipm = BatchedIPM(eng, max_iter=150)
    r = ipm.solve(eng.get_starting_point()[None, :])
    assert r["status"][0] in (0, 1, 2, 3)
    if r["status"][0] in (0, 1):
        assert r["kkt_error"][0] <= (1e-8 if r["status"][0] == 0 else 1e-6)
        assert abs(-r["obj"][0] * 301454.0 - 7529.71) < 0.05
    else:
        assert r["kkt_error"][0] > 1e-8 and np.isfinite(r["x"]).all()
    ipm.close()
    eng.close()


@pytest.mark.gpu
def test_reference_hypersensitive_example_end_to_end_on_the_device(built):
    """example/hypersensitive/HyperSensitive.cpp:17-57 as shipped: tf = 5000, hessian-approximation=exact,
    first-derive=analytic, mesh-refine-methods=hp-Liu, max-grid-num=20.  Every mesh's NLP is solved by rpm_ipm_*; the loop
    ends by itself (NoMoreRefine) at the turnpike cost V(1.5) + W(1) of the infinite-horizon problem."""
    from scipy.integrate import quad
    from lpopc_amd.application import LpopcApplication, console_not_print
    V = quad(lambda x: -x ** 3 + np.sqrt(x ** 6 + x ** 2), 0, 1.5)[0]
    W = quad(lambda x: x ** 3 + np.sqrt(x ** 6 + x ** 2), 0, 1.0)[0]
    app = LpopcApplication(console_not_print)
    app.SetOptimalControlProblem(problems.hypersensitive())
    app.Options().SetStringValue("hessian-approximation", "exact")
    app.Options().SetStringValue("first-derive", "analytic")
    app.Options().SetStringValue("mesh-refine-methods", "hp-Liu")
    app.Options().SetIntegerValue("max-grid-num", 20)
    assert app.SolveOptimalProblem(device=0) is True
    assert abs(app.objective - (V + W)) < 1e-5
    assert 3 <= app.meshrefiner_.CurrentGrid() < 20


TINY = [("brachistochrone_1x10_config1", lambda: problems.brachistochrone(1, 10)), ("brachistochrone_1x3", lambda: problems.brachistochrone(1, 3)),
        ("bryson_denham_1x4", lambda: problems.bryson_denham(1, 4)), ("hypersensitive_1x3", lambda: problems.hypersensitive([-1, 1], [3], tf=10.0)),
        ("hypersensitive_ragged", lambda: problems.hypersensitive([-1, -0.9, 0.0, 0.2, 1], [3, 12, 2, 7], tf=40.0)),
        ("quadrotor_1x2", lambda: problems.quadrotor(1, 2))]


@pytest.mark.gpu
@pytest.mark.parametrize("name,make", TINY, ids=[c[0] for c in TINY])
def test_tiny_and_ragged_layouts(built, name, make):
    """Bands narrower than a 16-column block, a band shorter than its width, ragged hp meshes, BASELINE config 1."""
    from lpopc_amd.engine import BatchedIPM, NLPEngine
    prob = make()
    eng = NLPEngine(prob, _exact(), n_instances=2, device=0)
    o = orc.Oracle(prob, _exact())
    ipm = BatchedIPM(eng, max_iter=200)
    r = ipm.solve(np.tile(o.starting_point(), (2, 1)))
    ref = ipm_oracle.solve(o, o.starting_point(), max_iter=200)
    assert (r["status"] == 0).all() and ref["status"] == 0
    assert (np.abs(r["iterations"] - ref["iterations"]) <= 1).all()
    assert np.max(np.abs(r["obj"] - ref["obj"])) <= 1e-8 * max(1.0, abs(ref["obj"]))
    assert np.array_equal(r["x"][0], r["x"][1])                 # identical instances take identical paths
    ipm.close()
    eng.close()


def test_restatement_restoration_rescues_bryson_denham_on_the_default_mesh():
    """example/bryson-denham as shipped (default 10 x 4 mesh) under the monotone barrier rule: the filter line search gives up
    at an infeasible point (theta ~ 4) where Ipopt enters its restoration phase; the restoration phase brings it back.
    (With the adaptive rule, the default, this start needs no restoration.)"""
    o = orc.Oracle(problems.bryson_denham(), _exact())
    off = ipm_oracle.solve(o, o.starting_point(), tol=1e-6, resto=0, mu_strategy="monotone")
    on = ipm_oracle.solve(o, o.starting_point(), tol=1e-6, mu_strategy="monotone")
    assert ipm_oracle.solve(o, o.starting_point(), tol=1e-6)["restorations"] == 0
    assert off["status"] == 3 and on["status"] == 0 and on["restorations"] >= 1
    assert abs(on["obj"] - 4.0) < 5e-3                      # coarse mesh


@pytest.mark.gpu
def test_restoration_and_multiplier_passes_through_the_assembling_factor_kernel(built):
    """The restoration phase (mode 2: another matrix and right-hand side) and the least-squares-multiplier pass on leaving it
    (mode 3) with level 1 assembled and forward-substituted inside kkt_factor_dense_kernel: the same iterates, bit for bit, as with
    the fill kernel and the separate substitution (option fused_fill), and restorations do happen on this run."""
    from lpopc_amd.engine import BatchedIPM, NLPEngine
    prob = problems.bryson_denham()
    out = []
    for fused in (1, 0):
        eng = NLPEngine(prob, _exact(), n_instances=2, device=0)
        eng.set_option("ipm_nested", 1)
        ipm = BatchedIPM(eng, tol=1e-6, trace=300, mu_strategy="monotone")
        ipm.set_option("fused_fill", fused)
        x0 = np.tile(eng.get_starting_point(), (2, 1))
        x0[1] *= 1 + 1e-3 * np.random.RandomState(4).uniform(-1, 1, x0.shape[1])
        r = ipm.solve(x0)
        out.append((r, [ipm.trace(bi).copy() for bi in range(2)], ipm.restorations().copy()))
        ipm.close()
        eng.close()
    (a, ta, ra), (b, tb, rb) = out
    assert (a["status"] == 0).all() and (ra >= 1).all()
    assert np.array_equal(ra, rb) and np.array_equal(a["iterations"], b["iterations"]) and np.array_equal(a["x"], b["x"])
    assert all(np.array_equal(p, q) for p, q in zip(ta, tb))


@pytest.mark.gpu
def test_device_restoration_against_restatement(built):
    from lpopc_amd.engine import BatchedIPM, NLPEngine
    prob = problems.bryson_denham()
    eng = NLPEngine(prob, _exact(), n_instances=2, device=0)
    o = orc.Oracle(prob, _exact())
    ref = ipm_oracle.solve(o, o.starting_point(), tol=1e-6, mu_strategy="monotone")     # (the adaptive rule needs no restoration here)
    ipm = BatchedIPM(eng, tol=1e-6, trace=200, mu_strategy="monotone")
    r = ipm.solve(np.tile(o.starting_point(), (2, 1)))
    assert (r["status"] == 0).all() and ref["status"] == 0
    assert (ipm.restorations() == ref["restorations"]).all() and ref["restorations"] >= 1
    assert np.max(np.abs(r["obj"] - ref["obj"])) <= 1e-6 * abs(ref["obj"])
    # Before the line search gives up it crawls: steps of 1e-3 .. 1e-5 whose acceptance is decided at rounding level, so the two
    # paths enter the restoration a few iterations apart (14 / 19 when this was written).  From there on they agree step by
    # step: the same number of restoration iterations, the same number of regular iterations after it.
    ls_ref = np.array([t["ls"] for t in ref["trace"]])
    for bi in range(2):
        ls_dev = ipm.trace(bi)[:, 7]
        assert (ls_dev < 0).sum() == (ls_ref < 0).sum() >= 1
        after_dev, after_ref = len(ls_dev) - 1 - np.nonzero(ls_dev < 0)[0][-1], len(ls_ref) - 1 - np.nonzero(ls_ref < 0)[0][-1]
        assert abs(int(after_dev) - int(after_ref)) <= 1
    assert (np.abs(r["iterations"] - ref["iterations"]) <= 8).all()
    ipm.set_option("restoration", 0)                         # without it: the verdict Ipopt-less code has to give
    r0 = ipm.solve(np.tile(o.starting_point(), (2, 1)))
    assert (r0["status"] == 3).all()
    ipm.close()
    eng.close()


@pytest.mark.gpu
def test_instances_of_a_batch_switch_modes_independently(built):
    """One batch, three starts: lpopc's guess (the line search gives up -> restoration phase -> least-squares multipliers), the
    converged solution (a few regular iterations), a perturbed guess.  Every kernel takes the mode per instance; the solve loop
    on the host does not know about modes at all."""
    from lpopc_amd.engine import BatchedIPM, NLPEngine
    prob = problems.bryson_denham()
    o = orc.Oracle(prob, _exact())
    ref = ipm_oracle.solve(o, o.starting_point(), tol=1e-6, mu_strategy="monotone")
    assert ref["status"] == 0 and ref["restorations"] >= 1
    x0 = np.stack([o.starting_point(), ref["x"], o.starting_point() * (1 + 1e-2 * np.random.RandomState(2).uniform(-1, 1, o.n))])
    eng = NLPEngine(prob, _exact(), n_instances=3, device=0)
    ipm = BatchedIPM(eng, tol=1e-6, mu_strategy="monotone")
    r = ipm.solve(x0)
    assert (r["status"] == 0).all(), r["status"]
    assert np.max(np.abs(r["obj"] - ref["obj"])) <= 1e-5 * abs(ref["obj"])
    n_resto = ipm.restorations()
    assert n_resto[0] == ref["restorations"] and n_resto[1] == 0
    assert r["iterations"][1] < r["iterations"][0] - 5
    for bi in (1, 2):                                     # each against the restatement run from its own start
        rb = ipm_oracle.solve(o, x0[bi], tol=1e-6, mu_strategy="monotone")
        assert rb["status"] == 0 and abs(r["obj"][bi] - rb["obj"]) <= 1e-5 * abs(rb["obj"])
    ipm.close()
    eng.close()


@pytest.mark.gpu
def test_sweep_with_per_instance_targets(built):
    """An MPC sweep whose instances differ in the problem constants (tracking target): the device solver against the
    restatement run on separately built problems; the batched Hessian uses every instance's own constants too."""
    from lpopc_amd.engine import BatchedIPM, NLPEngine
    B = 4
    rng = np.random.RandomState(3)
    prefs = [tuple(rng.uniform(-1.0, 1.0, size=3)) for _ in range(B)]
    probs = [problems.quadrotor(2, 4, pref=p) for p in prefs]
    eng = NLPEngine(probs[0], _exact(), n_instances=B, device=0)
    for b in range(1, B):
        eng.set_instance_constants(b, probs[b].GetOpimalProblemFuns().consts)
    ipm = BatchedIPM(eng)
    # the same start for everyone (the guess of instance 0; the guesses differ only through the target)
    x0 = np.tile(orc.Oracle(probs[0], _exact()).starting_point(), (B, 1))
    r = ipm.solve(x0)
    for b in range(B):
        ref = ipm_oracle.solve(orc.Oracle(probs[b], _exact()), x0[b])
        assert r["status"][b] == ref["status"] == 0
        assert abs(int(r["iterations"][b]) - ref["iterations"]) <= 1
        assert abs(r["obj"][b] - ref["obj"]) <= 1e-8 * max(1.0, abs(ref["obj"]))
    assert len(set(np.round(r["obj"], 6))) == B
    ipm.close()
    eng.close()


@pytest.mark.gpu
def test_all_bounds_equals_per_instance_bounds_and_warm_start(built):
    """rpm_ipm_set_all_bounds == B calls of rpm_ipm_set_bounds; a solve restarted from its own solution with warm-start
    options (small mu_init, no bound push) needs fewer iterations and lands on the same optimum."""
    from lpopc_amd.engine import BatchedIPM, NLPEngine, RpmError
    B = 3
    prob = problems.quadrotor(2, 4)
    eng = NLPEngine(prob, _exact(), n_instances=B, device=0)
    xl, xu, _, _ = eng.get_bounds_info()
    N1 = 2 * 4 + 1
    idx = [i * N1 for i in range(12)]
    rng = np.random.RandomState(8)
    XL, XU = np.tile(xl, (B, 1)), np.tile(xu, (B, 1))
    XL[:, idx] = XU[:, idx] = rng.uniform(-0.2, 0.2, size=(B, 12))
    x0 = np.tile(eng.get_starting_point()[:eng.n], (B, 1))
    a = BatchedIPM(eng)
    for b in range(B):
        a.set_bounds(b, XL[b], XU[b])
    ra = a.solve(x0)
    a.close()
    c = BatchedIPM(eng)
    c.set_all_bounds(XL, XU)
    rc = c.solve(x0)
    assert np.array_equal(ra["x"], rc["x"]) and np.array_equal(ra["iterations"], rc["iterations"])
    for k, v in (("mu_init", 1e-6), ("bound_push", 1e-9), ("bound_frac", 1e-9)):
        c.set_option(k, v)
    rw = c.solve(rc["x"])
    assert (rw["status"] == 0).all() and (rw["iterations"] < rc["iterations"]).all()
    assert np.max(np.abs(rw["obj"] - rc["obj"])) <= 1e-7 * np.max(np.abs(rc["obj"]))
    bad = XL.copy()
    free = np.nonzero(xl != xu)[0][0]
    bad[1, free] = XU[1, free]
    with pytest.raises(RpmError):
        c.set_all_bounds(bad, XU)
    c.close()
    eng.close()


# ---- committed regression fixtures (tests/golden/make_ipm_golden.py) -----------------------------------------------------
def _ipm_fixtures():
    import glob
    import os
    return sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ipm", "*.npz")))


def _ipm_case(path):
    import importlib.util
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    spec = importlib.util.spec_from_file_location("make_ipm_golden", os.path.join(here, "golden", "make_ipm_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    make, opts = mg.CASES[os.path.basename(path)[:-4]]
    return make(), opts


@pytest.mark.parametrize("path", _ipm_fixtures(), ids=lambda p: p.split("/")[-1])
def test_restatement_reproduces_ipm_fixture(path):
    prob, opts = _ipm_case(path)
    z = np.load(path)
    if "launch_4x8" in path:           # two minutes of dense linear algebra: regenerated by make_ipm_golden.py only; what it holds is checked
        assert z["status"][0] in (0, 1) and abs(-z["obj"][0] * 301454.0 - 7529.71) < 0.01
        return
    o = orc.Oracle(prob, _exact())
    r = ipm_oracle.solve(o, z["x0"], **opts)
    # LAPACK's threading changes rounding from run to run: one iteration either way on the long Delta-III solve, 1e-7 in the
    # weakly determined directions of x
    assert r["status"] == z["status"][0] and abs(r["iterations"] - z["iterations"][0]) <= 2 and r["restorations"] == z["restorations"][0]
    assert np.allclose(r["x"], z["x"], rtol=0, atol=1e-7) and abs(r["obj"] - z["obj"][0]) <= 1e-9 * max(1.0, abs(z["obj"][0]))
    if "launch" in path:     # the published optimum of the Delta-III problem: 7529.71 kg of final mass (mass unit = lift-off mass)
        assert r["status"] in (0, 1) and abs(-r["obj"] * 301454.0 - 7529.71) < 0.01


@pytest.mark.gpu
@pytest.mark.parametrize("path", _ipm_fixtures(), ids=lambda p: p.split("/")[-1])
def test_device_reproduces_ipm_fixture(built, path):
    from lpopc_amd.engine import BatchedIPM, NLPEngine
    prob, opts = _ipm_case(path)
    z = np.load(path)
    eng = NLPEngine(prob, _exact(), device=0)
    ipm = BatchedIPM(eng, **opts)
    r = ipm.solve(z["x0"][None, :])
    if "launch" in path:
        # Delta-III from lpopc's default guess: hundreds of iterations through inertia corrections of 1e5 and (depending on
        # rounding) the restoration phase — the paths differ, the optimum does not.  Status 1 = Ipopt's "acceptable level":
        # the iterate sits on a bound that bound_relax_factor moved out by 1e-8, which leaves 2e-8 of infeasibility.
        assert r["status"][0] in (0, 1) and z["status"][0] in (0, 1)
        assert abs(r["obj"][0] - z["obj"][0]) <= 1e-8
        assert abs(-r["obj"][0] * 301454.0 - 7529.71) < 0.01           # kg of final mass: the published optimum of the problem
    else:
        assert r["status"][0] == z["status"][0] and ipm.restorations()[0] == z["restorations"][0]
        assert abs(int(r["iterations"][0]) - int(z["iterations"][0])) <= (8 if z["restorations"][0] else 2)   # see test_device_restoration_against_restatement
        assert abs(r["obj"][0] - z["obj"][0]) <= 1e-6 * max(1.0, abs(z["obj"][0]))
        assert np.max(np.abs(r["x"][0] - z["x"])) <= 1e-4 * max(1.0, np.max(np.abs(z["x"])))
    ipm.close()
    eng.close()


ADAPTIVE = [("bryson_denham_default", lambda: problems.bryson_denham(), 1e-8), ("brachistochrone", lambda: problems.brachistochrone(2, 10), 1e-8),
            ("quadrotor_3x6", lambda: problems.quadrotor(3, 6, pref=(0.4, 0.8, -0.6)), 1e-8),
            ("hypersensitive", lambda: problems.hypersensitive(np.linspace(-1, 1, 7).tolist(), [10] * 6, tf=30.0), 1e-8)]


@pytest.mark.parametrize("name,make,tol", ADAPTIVE, ids=[c[0] for c in ADAPTIVE])
def test_restatement_adaptive_barrier_update(name, make, tol):
    """mu_strategy = adaptive (the default: what the reference asks Ipopt for; LOQO oracle, kkt-error globalisation) ends where
    the monotone rule ends on the convex / well-behaved problems, in no more than 1.5 x the iterations."""
    o = orc.Oracle(make(), _exact())
    a = ipm_oracle.solve(o, o.starting_point(), tol=tol, mu_strategy="adaptive")
    b = ipm_oracle.solve(o, o.starting_point(), tol=tol, mu_strategy="monotone")
    assert a["status"] == 0 and b["status"] == 0
    assert abs(a["obj"] - b["obj"]) <= 1e-6 * max(1.0, abs(b["obj"]))
    assert a["iterations"] <= 1.5 * b["iterations"] + 2
    assert len(set(round(np.log10(t["mu"]), 3) for t in a["trace"])) > len(set(round(np.log10(t["mu"]), 3) for t in b["trace"]))   # mu moves every iteration


@pytest.mark.gpu
@pytest.mark.parametrize("name,make,tol", ADAPTIVE, ids=[c[0] for c in ADAPTIVE])
def test_device_adaptive_barrier_update_against_restatement(built, name, make, tol):
    from lpopc_amd.engine import BatchedIPM, NLPEngine
    prob = make()
    o = orc.Oracle(prob, _exact())
    ref = ipm_oracle.solve(o, o.starting_point(), tol=tol, mu_strategy="adaptive")
    eng = NLPEngine(prob, _exact(), n_instances=2, device=0)
    ipm = BatchedIPM(eng, tol=tol, mu_strategy=1, trace=100)
    r = ipm.solve(np.tile(o.starting_point(), (2, 1)))
    assert (r["status"] == 0).all() and ref["status"] == 0
    assert (np.abs(r["iterations"].astype(int) - ref["iterations"]) <= 1).all()
    assert np.max(np.abs(r["obj"] - ref["obj"])) <= 1e-8 * max(1.0, abs(ref["obj"]))
    tr = ipm.trace(0)
    # the same barrier parameters from the oracle's rule, step by step (from the second step on the constraint Hessians — second
    # differences that differ by 1e-9 .. 1e-3 between CPU and GPU — enter, and the two paths run ~1e-4 apart to the same optimum)
    for k in range(min(4, len(tr), len(ref["trace"]))):
        assert abs(tr[k, 2] - ref["trace"][k]["mu"]) <= (1e-6 if k < 2 else 1e-2) * ref["trace"][k]["mu"], (k, tr[k, 2], ref["trace"][k]["mu"])
    ipm.close()
    eng.close()


@pytest.mark.gpu
def test_device_solves_delta_iii_with_the_adaptive_update(built):
    from lpopc_amd.engine import BatchedIPM, NLPEngine
    eng = NLPEngine(problems.launch(4, 8), _exact(), device=0)
    ipm = BatchedIPM(eng, max_iter=1500, mu_strategy=1)
    r = ipm.solve(eng.get_starting_point()[None, :])
    assert r["status"][0] in (0, 1), (r["status"], r["iterations"], r["kkt_error"])
    assert abs(-r["obj"][0] * 301454.0 - 7529.71) < 0.01
    ipm.close()
    eng.close()


@pytest.mark.gpu
def test_reference_launch_example_with_hp_liu_refinement_on_the_device(built):
    """example/launch/Launch.cpp (Delta-III on its own 4 x 1 x 20 mesh and guess) through the reference's outer loop with
    mesh-refine-methods=hp-Liu: every grid's NLP solved by rpm_ipm_*, error estimate and refinement on the GPU; the refined
    meshes have different node totals per phase, so the link-Hessian's mis-indexed entries are promoted to the border."""
    from lpopc_amd.application import LpopcApplication, console_not_print
    app = LpopcApplication(console_not_print)
    prob = problems.launch()
    app.SetOptimalControlProblem(prob)
    app.Options().SetStringValue("hessian-approximation", "exact")
    app.Options().SetStringValue("mesh-refine-methods", "hp-Liu")
    app.Options().SetIntegerValue("max-grid-num", 8)
    app.SolveOptimalProblem()                       # raises if a solve fails or the grid limit is hit
    assert app.meshrefiner_.CurrentGrid() >= 1
    assert abs(-app.objective * 301454.0 - 7529.71) < 0.06
    totals = [int(np.sum(prob.GetPhase(i).GetNodesPerInterval())) for i in range(4)]
    assert len(set(totals)) > 1, totals


@pytest.mark.gpu
def test_delta_iii_mesh_ladder_with_the_recorded_retry(built):
    """Delta-III from lpopc's default guess on a ladder of meshes through DeviceIPMSolver (NLPSolver::SolveNlp on the device):
    every mesh ends at the published optimum.  The path is chaotic and the problem degenerate: with the
    earlier constraint regularisation delta_c = 1e-8 the 4 x 8 x 8 mesh stalled (status 3) and was solved by the solver's
    recorded retry with bound_relax_factor 1e-7; with delta_c = 1e-9, the default since, no mesh of the ladder needs it."""
    from lpopc_amd.application import DeviceIPMSolver
    from lpopc_amd.engine import NLPEngine
    retried = 0
    for K, Nk in ((2, 6), (4, 8), (8, 8), (16, 8)):
        eng = NLPEngine(problems.launch(K, Nk), _exact(), device=0)
        solver = DeviceIPMSolver(tol=1e-8, maxiter=3000)
        assert solver.SolveNlp(eng), (K, Nk, solver.attempts)
        _, _, obj = eng.get_solution()
        assert int(solver.last["status"][0]) in (0, 1) and abs(-obj * 301454.0 - 7529.71) < 0.01, (K, Nk, obj, solver.attempts)
        retried += len(solver.attempts) - 1
        eng.close()
    assert retried <= 2


@pytest.mark.gpu
def test_large_instances_in_a_batch_use_several_workgroups_each(built):
    """A few large instances (n >= 4096, at most 32 of them): the vector kernels run several workgroups per instance and the last
    one to arrive combines their partial sums in a fixed order (vec_combine).  Three instances from the same start, one of them
    perturbed: the unperturbed two must agree bit for bit with each other and with a one-instance solve."""
    from lpopc_amd.engine import BatchedIPM, NLPEngine
    prob = problems.launch(16, 8)
    one = NLPEngine(prob, _exact(), device=0)
    x0 = one.get_starting_point()
    assert one.n >= 4096
    ipm1 = BatchedIPM(one, max_iter=60)
    r1 = ipm1.solve(x0[None, :])
    ipm1.close()
    one.close()
    eng = NLPEngine(prob, _exact(), n_instances=3, device=0)
    ipm = BatchedIPM(eng, max_iter=60)
    xs = np.stack([x0, x0 * (1 + 1e-3 * np.random.RandomState(0).uniform(-1, 1, x0.size)), x0])
    r = ipm.solve(xs)
    assert np.array_equal(r["x"][0], r["x"][2]) and np.array_equal(r["x"][0], r1["x"][0])
    assert r["iterations"][0] == r["iterations"][2] == r1["iterations"][0] == 60 and r["obj"][0] == r1["obj"][0]
    assert not np.array_equal(r["x"][0], r["x"][1])
    ipm.close()
    eng.close()


@pytest.mark.gpu
def test_device_solves_the_metric_problem(built):
    """BASELINE's metric problem at full size — Delta-III, 4 phases x 64 intervals x 16 LGR points, n = 40 996, KKT order
    73 801 — from lpopc's default guess (example/launch/Launch.cpp:200-457), on the device: nested dissection over the 256 mesh
    intervals, bound relaxation, second-order corrections, restoration phase.  No CPU run to compare with at this size (the
    dense restatement would need a 44 GB matrix); the check is the problem's published optimum, 7529.71 kg of final mass (the
    finer mesh ends 0.04 kg below the 4 x 8 mesh's value: finite-difference noise in a very flat optimum)."""
    from lpopc_amd.engine import BatchedIPM, NLPEngine
    eng = NLPEngine(problems.launch(64, 16), _exact(), device=0)
    ipm = BatchedIPM(eng, max_iter=2000)
    r = ipm.solve(eng.get_starting_point()[None, :])
    assert r["status"][0] in (0, 1), (r["status"], r["iterations"], r["kkt_error"])
    assert r["kkt_error"][0] <= 1e-6
    assert abs(-r["obj"][0] * 301454.0 - 7529.71) < 0.1
    g = orc.Oracle(problems.launch(64, 16)).eval_g(r["x"][0])
    _, _, gl, gu = orc.Oracle(problems.launch(64, 16)).bounds()
    assert max((gl - g).max(), (g - gu).max()) < 1e-7
    assert ipm.subproblems().shape[0] == 256 + 11 + 1          # interval blocks, groups of the separator system, last level
    ipm.close()
    eng.close()

"""CPU-side tests of the product: the C-ABI library loads and exports every symbol the header
declares, the host set-up (layout, bounds, guess, tables, structure) equals the oracle bit for
bit, errors are reported the way the reference's checkers report them, and every evaluation fails
loudly without a GPU (there is no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from lpopc_amd import problems
from lpopc_amd.engine import ABI_SYMBOLS, NLPEngine, RpmError, lib
from lpopc_amd.problem import (Linkage, LpopcException, OptimalProblem, Options, Phase, ProblemFunctor,
                               apply_mesh_defaults)
from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(built):
    header = open(os.path.join(ROOT, "include", "rpm_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|void|const char\*|rpm_engine\*|rpm_ipm\*)\s+(rpm_[a-z0-9_]+)\s*\(", header, re.M))
    assert declared == set(ABI_SYMBOLS), declared ^ set(ABI_SYMBOLS)
    L = C.CDLL(built)
    for s in declared:
        assert hasattr(L, s), s


CONFIGS = ["brachistochrone", "bryson_denham", "launch_default", "climb", "launch", "hypersensitive", "quadrotor"]


@pytest.mark.parametrize("name", CONFIGS)
def test_host_setup_equals_oracle_bit_for_bit(built, name):
    prob = problems.config(name)
    o, e = orc.Oracle(prob), NLPEngine(prob)
    assert (e.n, e.m, e.nnz_jac) == (o.n, o.m, o.nnz_jac)
    for a, b in zip(e.get_bounds_info(), o.bounds()):
        assert np.array_equal(a, b)
    assert np.array_equal(e.get_starting_point(), o.starting_point())
    i1, j1 = e.eval_jac_g_structure()
    i2, j2 = o.jac_structure()
    assert np.array_equal(i1, i2) and np.array_equal(j1, j2)
    for ph in range(e.n_phases):
        t1, t2 = e.phase_tables(ph), o.phase_tables(ph)
        for k in t1:
            assert np.array_equal(t1[k], t2[k]), (ph, k)
    assert e.get_nlp_info()[4] == 0  # C_STYLE


def test_ragged_mesh_tiles_cover_every_node(built):
    p = problems.launch()
    for i, (mesh, nodes) in enumerate([([-1, -0.6, 0.1, 1], [5, 23, 2]), ([-1, 0.5, 1], [16, 17]),
                                       ([-1, 1], [33]), ([-1, -0.9, -0.5, 0.0, 0.25, 1], [3, 4, 7, 12, 16])]):
        problems.set_mesh(p.GetPhase(i), mesh, nodes)
    for T in (16, 32, 64):
        e = NLPEngine(p, tile_nodes=T)
        assert e.get_option("tile_nodes") == T
        o = orc.Oracle(p)
        assert np.array_equal(e.eval_jac_g_structure()[0], o.jac_structure()[0])


def test_solver_layout_options_round_trip(built):
    """The engine options the device solver's layout reads at rpm_ipm_create (host side only): defaults, set / get, bad values."""
    e = NLPEngine(problems.brachistochrone())
    assert e.get_option("ipm_local_border") == 1 and e.get_option("ipm_nested_group") == 0
    e.set_option("ipm_local_border", 0)
    e.set_option("ipm_nested_group", 48)
    assert e.get_option("ipm_local_border") == 0 and e.get_option("ipm_nested_group") == 48
    with pytest.raises(RpmError):
        e.set_option("ipm_nested_group", -1)


def test_no_gpu_means_loud_failure_not_fallback(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    e = NLPEngine(problems.brachistochrone())
    x = e.get_starting_point()
    for call in (lambda: e.eval_g(x), lambda: e.eval_jac_g(x), lambda: e.eval_f(x), lambda: e.eval_grad_f(x),
                 lambda: e.device_init(0)):
        with pytest.raises(RpmError) as ei:
            call()
        assert ei.value.code == 3 and "no CPU fallback" in str(ei.value)


def test_error_reporting_matches_reference_checkers(built):
    # inconsistent bounds (LpBoundsChecker.cpp:77-84)
    p = problems.brachistochrone()
    p.GetPhase(0).GetstateMax()[0].state[1] = -5.0
    with pytest.raises(RpmError) as ei:
        NLPEngine(p)
    assert "Bounds on State are Inconsistent" in str(ei.value)
    # mesh must span -1..1 (LpMeshRefiner.cpp:40-46)
    p = problems.brachistochrone(None, None)
    p.GetPhase(0).SetMeshPoints(-1)
    p.GetPhase(0).SetMeshPoints(0.9)
    with pytest.raises(LpopcException) as ei:
        NLPEngine(p)
    assert "meshPoints must span -1 to +1" in str(ei.value)
    # guess needs two distinct time knots (LpGuessChecker.cpp:39-63)
    p = problems.brachistochrone()
    p.GetPhase(0).vtimeguess[1] = p.GetPhase(0).vtimeguess[0]
    with pytest.raises(RpmError) as ei:
        NLPEngine(p)
    assert "unique values" in str(ei.value)
    # dimensions must match the compiled functor
    ph = Phase(1, 2, 1, 0, 0, 0)
    with pytest.raises(Exception):
        op = OptimalProblem(1, 0, ProblemFunctor(problems.RPM_PROBLEM_BRACHISTOCHRONE, [9.8]))
        op.AddPhase(ph)
        NLPEngine(op)
    # analytic mode only where the functor ships derivatives (Launch does not: Launch.cpp:648-650)
    o = Options()
    o.SetStringValue("first-derive", "analytic")
    with pytest.raises(RpmError) as ei:
        NLPEngine(problems.launch(2, 4), o)
    assert ei.value.code == 2
    # eval_h on an engine created with limited-memory Hessian (the reference default): reported, never ignored
    e = NLPEngine(problems.brachistochrone())
    with pytest.raises(RpmError) as ei:
        e.eval_h(e.get_starting_point(), 1.0, np.zeros(e.m))
    assert ei.value.code == 2
    # structure/values protocol argument checks
    L = lib()
    assert L.rpm_eval_jac_g(e._h, e.n, None, 0, e.m, e.nnz_jac + 1, None, None, None) == 1
    assert L.rpm_get_starting_point(e._h, e.n, 0, None, 0, None, None, e.m, 0, None) == 1


def test_finalize_solution_round_trip(built):
    e = NLPEngine(problems.bryson_denham())
    x, lam = np.arange(e.n, dtype=float), np.arange(e.m, dtype=float) * 0.5
    e.finalize_solution(0, x, lam, -3.25)
    xs, ls, obj = e.get_solution()
    assert np.array_equal(xs, x) and np.array_equal(ls, lam) and obj == -3.25


def test_mirrored_setup_api_semantics():
    # Options: 13 registered options with the reference defaults and its validation
    o = Options()
    assert o.GetNumericValue("finite-difference-tol") == 1e-6 and o.GetStringValue("first-derive") == "finite-difference"
    assert o.GetIntegerValue("Nmax") == 16 and o.GetStringValue("hessian-approximation") == "limited-memory"
    with pytest.raises(LpopcException):
        o.SetStringValue("first-derive", "symbolic")
    with pytest.raises(LpopcException):
        o.SetNumericValue("no-such-option", 1.0)
    # SetStateGuess is 1-based and appends per state (LpOptimalProblem.hpp:135-143)
    ph = Phase(1, 2, 1, 0, 0, 0)
    ph.SetStateGuess(1, 0.0)
    ph.SetStateGuess(1, 1.0)
    ph.SetStateGuess(2, 5.0)
    assert ph.GetStateGuess() == [[0.0, 1.0], [5.0]]
    lk = Linkage(1, 2, 3)
    assert (lk.LeftPhase(), lk.RightPhase()) == (1, 2)             # 0-based accessors (:264-269)
    # default mesh: one interval [-1,1] with 20 nodes (SURVEY B-16)
    assert apply_mesh_defaults(Phase(1, 1, 1, 0, 0, 0)) == ([-1.0, 1.0], [20])
    op = OptimalProblem(1, 0, ProblemFunctor(2))
    with pytest.raises(LpopcException):
        op.GetPhase(0)


def test_interval_shard_segments_partition_the_vectors(built):
    prob = problems.launch(8, 4)
    for world in (2, 3, 8):
        engs = [NLPEngine(prob, shard_mode=1, shard_rank=r, shard_world=world) for r in range(world)]
        for which, size in ((0, engs[0].m), (1, engs[0].nnz_jac)):
            cover = np.zeros(size, dtype=np.int32)
            for r in range(world):
                segs, plen = engs[0].shard_segments(which, r)
                assert plen == sum(s[1] for s in segs)
                pos = 0
                for off, ln, p in segs:
                    assert p == pos
                    pos += ln
                    cover[off:off + ln] += 1
            assert np.all(cover == 1)                                # every entry owned by exactly one rank


def test_auto_scale_is_refused_loudly(built):
    opts = Options()
    opts.SetStringValue("auto-scale", "yes")
    with pytest.raises(LpopcException):
        NLPEngine(problems.bryson_denham(), opts)



def test_instance_constants_host_validation(built):
    """rpm_set_instance_constants: count and instance are validated on the host (no device needed); interval-sharded
    engines refuse it."""
    from lpopc_amd.engine import NLPEngine, RpmError
    prob = problems.quadrotor(2, 3)
    consts = prob.GetOpimalProblemFuns().consts
    eng = NLPEngine(prob, n_instances=3)
    eng.set_instance_constants(2, consts)
    for bad in (lambda: eng.set_instance_constants(3, consts), lambda: eng.set_instance_constants(-1, consts),
                lambda: eng.set_instance_constants(0, consts[:-1])):
        with pytest.raises(RpmError):
            bad()
    eng.close()
    sharded = NLPEngine(problems.launch(8, 4), shard_mode=1, shard_rank=0, shard_world=2)
    with pytest.raises(RpmError):
        sharded.set_instance_constants(0, problems.launch(8, 4).GetOpimalProblemFuns().consts)
    sharded.close()

"""ctypes mirror of the C-ABI structs in include/rpm_hip.h and the lowering of a Python
OptimalProblem (lpopc_amd.problem) to an rpm_problem_desc.

Nothing here computes anything: it is the wire format between the host-side mirror of the
reference's set-up API (Core/LpOptimalProblem.hpp:30-326) and the native library.
"""
import ctypes as C

import numpy as np

RPM_ABI_VERSION = 2

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)


class rpm_phase_desc(C.Structure):
    _fields_ = [
        ("nx", C.c_int), ("nu", C.c_int), ("nq", C.c_int), ("nc", C.c_int), ("ne", C.c_int),
        ("n_intervals", C.c_int),
        ("mesh_points", c_double_p),
        ("nodes_per_interval", c_int_p),
        ("t0_min", C.c_double), ("tf_min", C.c_double),
        ("t0_max", C.c_double), ("tf_max", C.c_double),
        ("state_min", c_double_p), ("state_max", c_double_p),
        ("control_min", c_double_p), ("control_max", c_double_p),
        ("parameter_min", c_double_p), ("parameter_max", c_double_p),
        ("path_min", c_double_p), ("path_max", c_double_p),
        ("event_min", c_double_p), ("event_max", c_double_p),
        ("has_duration", C.c_int),
        ("duration_min", C.c_double), ("duration_max", C.c_double),
        ("n_guess", C.c_int),
        ("time_guess", c_double_p),
        ("state_guess", c_double_p),
        ("control_guess", c_double_p),
        ("parameter_guess", c_double_p),
    ]


class rpm_link_desc(C.Structure):
    _fields_ = [
        ("left_phase", C.c_int), ("right_phase", C.c_int),
        ("n_links", C.c_int),
        ("link_min", c_double_p), ("link_max", c_double_p),
    ]


class rpm_problem_desc(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int),
        ("problem_id", C.c_int),
        ("n_phases", C.c_int),
        ("phases", C.POINTER(rpm_phase_desc)),
        ("n_links", C.c_int),
        ("links", C.POINTER(rpm_link_desc)),
        ("n_consts", C.c_int),
        ("consts", c_double_p),
        ("fd_tol", C.c_double),
        ("first_derive", C.c_int),
        ("hessian_approximation", C.c_int),
        ("n_instances", C.c_int),
        ("shard_mode", C.c_int),
        ("shard_rank", C.c_int),
        ("shard_world", C.c_int),
    ]


class rpm_segment(C.Structure):
    _fields_ = [("off", C.c_int), ("len", C.c_int), ("pos", C.c_int)]


def _darr(values, keep):
    a = np.ascontiguousarray(np.asarray(values, dtype=np.float64).ravel())
    keep.append(a)
    return a.ctypes.data_as(c_double_p)


def _iarr(values, keep):
    a = np.ascontiguousarray(np.asarray(values, dtype=np.int32).ravel())
    keep.append(a)
    return a.ctypes.data_as(c_int_p)


def lower(problem, options=None, n_instances=1, shard_mode=0, shard_rank=0, shard_world=1):
    """OptimalProblem + Options -> (rpm_problem_desc, keepalive list)."""
    from .problem import Options, apply_mesh_defaults

    options = options or Options()
    if options.GetStringValue("auto-scale") == "yes":
        # LpScaleOCP (Core/LpSacleOCP.cpp) derives its function scales from random samples (CalculateFunScaleFromRand) and
        # the reference itself warns "auto-scale may fail" (Core/LpLpopcAlgorithm.cpp:295); it is out of this path's scope.
        from .problem import LpopcException
        raise LpopcException("auto-scale=yes is not supported by the GPU path (SURVEY scope: scaling stays off)")
    keep = []
    phases = (rpm_phase_desc * problem.GetPhaseNum())()
    for i in range(problem.GetPhaseNum()):
        ph = problem.GetPhase(i)
        mesh, nodes = apply_mesh_defaults(ph)
        d = phases[i]
        d.nx, d.nu, d.nq, d.nc, d.ne = ph.get_optimal_info()
        d.n_intervals = len(nodes)
        d.mesh_points = _darr(mesh, keep)
        d.nodes_per_interval = _iarr(nodes, keep)
        d.t0_min, d.tf_min = ph.GetTimeMin()
        d.t0_max, d.tf_max = ph.GetTimeMax()
        d.state_min = _darr([v for lim in ph.GetstateMin() for v in lim.state], keep)
        d.state_max = _darr([v for lim in ph.GetstateMax() for v in lim.state], keep)
        d.control_min = _darr(ph.GetcontrolMin(), keep)
        d.control_max = _darr(ph.GetcontrolMax(), keep)
        d.parameter_min = _darr(ph.GetparameterMin(), keep)
        d.parameter_max = _darr(ph.GetparameterMax(), keep)
        d.path_min = _darr(ph.GetpathMin(), keep)
        d.path_max = _darr(ph.GetpathMax(), keep)
        d.event_min = _darr(ph.GeteventMin(), keep)
        d.event_max = _darr(ph.GeteventMax(), keep)
        d.has_duration = 1 if ph.HasDuration() else 0
        if ph.HasDuration():
            d.duration_min, d.duration_max = ph.Getduration()
        tg = ph.GetTimeGuess()
        d.n_guess = len(tg)
        d.time_guess = _darr(tg, keep)
        d.state_guess = _darr([v for row in ph.GetStateGuess() for v in row], keep)
        d.control_guess = _darr([v for row in ph.GetControlGuess() for v in row], keep)
        d.parameter_guess = _darr(ph.GetparameterGuess(), keep)
    keep.append(phases)
    nl = problem.GetLinkageNum()
    links = (rpm_link_desc * max(nl, 1))()
    for i in range(nl):
        lk = problem.GetLinkage(i)
        links[i].left_phase = lk.LeftPhase() + 1
        links[i].right_phase = lk.RightPhase() + 1
        links[i].n_links = len(lk.GetLinkageMin())
        links[i].link_min = _darr(lk.GetLinkageMin(), keep)
        links[i].link_max = _darr(lk.GetLinkageMax(), keep)
    keep.append(links)
    fun = problem.GetOpimalProblemFuns()
    desc = rpm_problem_desc()
    desc.abi_version = RPM_ABI_VERSION
    desc.problem_id = fun.problem_id
    desc.n_phases = problem.GetPhaseNum()
    desc.phases = phases
    desc.n_links = nl
    desc.links = links
    desc.n_consts = len(fun.consts)
    desc.consts = _darr(fun.consts, keep)
    desc.fd_tol = float(options.GetNumericValue("finite-difference-tol"))
    desc.first_derive = 1 if options.GetStringValue("first-derive") == "analytic" else 0
    desc.hessian_approximation = 1 if options.GetStringValue("hessian-approximation") == "exact" else 0
    desc.n_instances = int(n_instances)
    desc.shard_mode = int(shard_mode)
    desc.shard_rank = int(shard_rank)
    desc.shard_world = int(shard_world)
    return desc, keep

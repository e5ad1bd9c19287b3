"""ctypes front-end of the CPU ORACLE (oracle/liborpm.so) — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
PARITY UNPINNED (see oracle/orpm.h): the reference ships no golden vectors and cannot be
built in this image, so this restatement is pinned by source citation and by mathematical
invariants only.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liborpm.so")
    srcs = [os.path.join(_HERE, f) for f in ("orpm_core.c", "orpm_problems.c", "orpm_hess.c", "orpm_post.c", "orpm_mesh.c", "orpm_hpliu.c", "orpm.h", "orpm_internal.h")]
    srcs.append(os.path.join(_HERE, "..", "include", "rpm_hip.h"))
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liborpm.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = build()   # mtime check against the C sources: a stale liborpm.so is rebuilt, never silently loaded
        L = C.CDLL(so)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
        L.orpm_create.restype = C.c_void_p
        L.orpm_create.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.orpm_destroy.argtypes = [C.c_void_p]
        L.orpm_set_cost_shape.argtypes = [C.c_void_p, C.c_int]
        L.orpm_set_cost_shape.restype = None
        L.orpm_get_nlp_info.argtypes = [C.c_void_p, ip, ip, ip, ip]
        L.orpm_get_bounds_info.argtypes = [C.c_void_p, dp, dp, dp, dp]
        L.orpm_get_starting_point.argtypes = [C.c_void_p, dp]
        L.orpm_eval_f.restype = C.c_double
        L.orpm_eval_f.argtypes = [C.c_void_p, dp]
        L.orpm_eval_grad_f.argtypes = [C.c_void_p, dp, dp]
        L.orpm_eval_g.argtypes = [C.c_void_p, dp, dp]
        L.orpm_jac_structure.argtypes = [C.c_void_p, ip, ip]
        L.orpm_eval_jac_g.argtypes = [C.c_void_p, dp, dp]
        L.orpm_hess_structure.argtypes = [C.c_void_p, ip, ip]
        L.orpm_eval_h.argtypes = [C.c_void_p, dp, C.c_double, dp, dp]
        L.orpm_get_phase_sizes.argtypes = [C.c_void_p, C.c_int, ip, ip, ip]
        L.orpm_get_phase_tables.argtypes = [C.c_void_p, C.c_int, dp, dp, ip, ip, dp, dp, ip, ip, dp]
        L.orpm_nlp2op.argtypes = [C.c_void_p, C.c_int, dp, dp, dp, dp, dp, dp, dp, dp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orpm_solution_error.argtypes = [C.c_void_p, C.c_int, dp, dp]
        L.orpm_ph_refine.argtypes = [C.c_void_p, C.c_int, dp, C.c_double, C.c_int, C.c_int, dp, ip, ip, dp]
        L.orpm_inverse.argtypes = [C.c_int, dp, dp]
        L.orpm_bary_tables.argtypes = [C.c_int, dp, C.c_int, dp, dp, dp, ip]
        L.orpm_hpliu_create.restype = C.c_void_p
        L.orpm_hpliu_create.argtypes = [C.c_int, C.c_double, C.c_int, C.c_double]
        L.orpm_hpliu_destroy.argtypes = [C.c_void_p]
        L.orpm_hpliu_refine.argtypes = [C.c_void_p, C.c_void_p, dp, C.c_int, dp, ip, ip, ip, ip]
        L.orpm_hpliu_alj.argtypes = [C.c_int, dp]
        L.orpm_lgr_points.argtypes = [C.c_int, dp, dp]
        L.orpm_colloc_d.argtypes = [C.c_int, dp, dp]
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def lgr_points(n):
    x, w = np.zeros(n), np.zeros(n)
    lib().orpm_lgr_points(n, _dp(x), _dp(w))
    return x, w


def colloc_d(points):
    pts = np.ascontiguousarray(points, dtype=np.float64)
    M = len(pts)
    D = np.zeros((M - 1) * M)
    lib().orpm_colloc_d(M, _dp(pts), _dp(D))
    return D.reshape((M - 1, M), order="F")


def hpliu_alj(N):
    a = np.zeros((N + 1) * (N + 1))
    lib().orpm_hpliu_alj(N, _dp(a))
    return a.reshape((N + 1, N + 1), order="F")


class HpLiu:
    """LiuHpMeshRefineAlg (Core/LpLiuHpMeshRefineAlg.cpp), stateful across meshes."""

    def __init__(self, n_phases, tol, nmax, ratio_r):
        self._h = lib().orpm_hpliu_create(n_phases, float(tol), int(nmax), float(ratio_r))
        self.P = n_phases

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orpm_hpliu_destroy(self._h)
            self._h = None

    def refine(self, oracle, x, cap=4096):
        """-> (no_more_refine, [(mesh_points, nodes_per_interval) per phase]); raises where the reference would throw."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        mesh, nodes = np.zeros(cap), np.zeros(cap, dtype=np.int32)
        moff, noff, nk = (np.zeros(self.P, dtype=np.int32) for _ in range(3))
        rc = lib().orpm_hpliu_refine(self._h, oracle._h, _dp(x), cap, _dp(mesh), _ip(nodes), _ip(moff), _ip(noff), _ip(nk))
        if rc < 0:
            raise RuntimeError("hp-Liu: the reference would throw here (empty find / row out of range / undefined cast)")
        return bool(rc), [(mesh[moff[p]:moff[p] + nk[p] + 1].copy(), nodes[noff[p]:noff[p] + nk[p]].copy()) for p in range(self.P)]


class Oracle:
    """CPU restatement of lpopc's NLPWrapper + LpopcIpopt for one OptimalProblem."""

    def __init__(self, problem, options=None):
        from lpopc_amd._abi import lower  # description structs only (the wire format)

        self._desc, self._keep = lower(problem, options)
        err = C.create_string_buffer(512)
        self._h = lib().orpm_create(C.byref(self._desc), err, 512)
        if not self._h:
            raise ValueError("oracle: " + err.value.decode())
        n, m, nj, nh = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        lib().orpm_get_nlp_info(self._h, C.byref(n), C.byref(m), C.byref(nj), C.byref(nh))
        self.n, self.m, self.nnz_jac, self.nnz_h = n.value, m.value, nj.value, nh.value
        self.n_phases = problem.GetPhaseNum()

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orpm_destroy(self._h)
            self._h = None

    def set_cost_shape(self, fair):
        """0: the reference's cost shape (COO product, per-call Find scans); 1: the fair shape (orpm.h).  Same numbers."""
        lib().orpm_set_cost_shape(self._h, 1 if fair else 0)

    def bounds(self):
        xl, xu, gl, gu = np.zeros(self.n), np.zeros(self.n), np.zeros(self.m), np.zeros(self.m)
        lib().orpm_get_bounds_info(self._h, _dp(xl), _dp(xu), _dp(gl), _dp(gu))
        return xl, xu, gl, gu

    def starting_point(self):
        x = np.zeros(self.n)
        lib().orpm_get_starting_point(self._h, _dp(x))
        return x

    def eval_f(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        return lib().orpm_eval_f(self._h, _dp(x))

    def eval_grad_f(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        g = np.zeros(self.n)
        lib().orpm_eval_grad_f(self._h, _dp(x), _dp(g))
        return g

    def eval_g(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        g = np.zeros(self.m)
        lib().orpm_eval_g(self._h, _dp(x), _dp(g))
        return g

    def jac_structure(self):
        i, j = np.zeros(self.nnz_jac, dtype=np.int32), np.zeros(self.nnz_jac, dtype=np.int32)
        lib().orpm_jac_structure(self._h, _ip(i), _ip(j))
        return i, j

    def eval_jac_g(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        v = np.zeros(self.nnz_jac)
        lib().orpm_eval_jac_g(self._h, _dp(x), _dp(v))
        return v

    def hess_structure(self):
        i, j = np.zeros(self.nnz_h, dtype=np.int32), np.zeros(self.nnz_h, dtype=np.int32)
        lib().orpm_hess_structure(self._h, _ip(i), _ip(j))
        return i, j

    def eval_h(self, x, obj_factor, lam):
        x = np.ascontiguousarray(x, dtype=np.float64)
        lam = np.ascontiguousarray(lam, dtype=np.float64)
        v = np.zeros(self.nnz_h)
        lib().orpm_eval_h(self._h, _dp(x), float(obj_factor), _dp(lam), _dp(v))
        return v

    def nlp2op(self, phase, x, lam):
        """Nlp2OpConverter::Nlp2OpControl for one phase -> dict of (N+1)-row column-major arrays."""
        d = self._desc.phases[phase]
        N = self.phase_tables(phase)["points"].size
        M = N + 1
        x = np.ascontiguousarray(x, dtype=np.float64)
        lam = np.ascontiguousarray(lam, dtype=np.float64)
        out = dict(time=np.zeros(M), state=np.zeros(M * d.nx), control=np.zeros(M * max(d.nu, 1)),
                   costate=np.zeros(M * d.nx), pathmult=np.zeros(M * max(d.nc, 1)), hamiltonian=np.zeros(M))
        mc, lc = C.c_double(), C.c_double()
        lib().orpm_nlp2op(self._h, phase, _dp(x), _dp(lam), _dp(out["time"]), _dp(out["state"]), _dp(out["control"]),
                          _dp(out["costate"]), _dp(out["pathmult"]), _dp(out["hamiltonian"]), C.byref(mc), C.byref(lc))
        out["control"] = out["control"][:M * d.nu]
        out["pathmult"] = out["pathmult"][:M * d.nc]
        out["mayer_cost"], out["lagrange_cost"] = mc.value, lc.value
        return out

    def solution_error(self, phase, x):
        """SolutionErrorChecker::CheckSolutionDiffError -> relative_error, (N+K+1) x nx."""
        d = self._desc.phases[phase]
        N = self.phase_tables(phase)["points"].size
        rows = N + d.n_intervals + 1
        x = np.ascontiguousarray(x, dtype=np.float64)
        rel = np.zeros(rows * d.nx)
        lib().orpm_solution_error(self._h, phase, _dp(x), _dp(rel))
        return rel.reshape((rows, d.nx), order="F")

    def ph_refine(self, phase, x, tol, nmin, nmax):
        """PhMeshRefineAlg::RefineMesh for one phase -> (no_more_refine, mesh_points, nodes_per_interval, emax)."""
        d = self._desc.phases[phase]
        K = d.n_intervals
        x = np.ascontiguousarray(x, dtype=np.float64)
        cap = K * 64 + 2
        mesh, nodes, nk, emax = np.zeros(cap * 8), np.zeros(cap * 8, dtype=np.int32), C.c_int(), np.zeros(K)
        done = lib().orpm_ph_refine(self._h, phase, _dp(x), float(tol), int(nmin), int(nmax), _dp(mesh), _ip(nodes),
                                    C.byref(nk), _dp(emax))
        return bool(done), mesh[:nk.value + 1].copy(), nodes[:nk.value].copy(), emax

    def phase_tables(self, phase):
        N, dn, on = C.c_int(), C.c_int(), C.c_int()
        lib().orpm_get_phase_sizes(self._h, phase, C.byref(N), C.byref(dn), C.byref(on))
        N, dn, on = N.value, dn.value, on.value
        t = dict(points=np.zeros(N), weights=np.zeros(N), d_rows=np.zeros(dn, dtype=np.int32),
                 d_cols=np.zeros(dn, dtype=np.int32), d_vals=np.zeros(dn), diag_vals=np.zeros(N),
                 doff_rows=np.zeros(on, dtype=np.int32), doff_cols=np.zeros(on, dtype=np.int32),
                 doff_vals=np.zeros(on))
        lib().orpm_get_phase_tables(self._h, phase, _dp(t["points"]), _dp(t["weights"]), _ip(t["d_rows"]),
                                    _ip(t["d_cols"]), _dp(t["d_vals"]), _dp(t["diag_vals"]),
                                    _ip(t["doff_rows"]), _ip(t["doff_cols"]), _dp(t["doff_vals"]))
        return t

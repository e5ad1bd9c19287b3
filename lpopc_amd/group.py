"""One process, several GPUs: rpm_group_* (include/rpm_hip.h) — the mesh intervals of ONE NLP sharded over the devices of
a node behind the C ABI, for the caller lpopc actually has: a single-process NLPSolver::SolveNlp driving one TNLP object
from one thread (Core/LpNLPSolver.cpp:13-53).  Method names follow LpopcIpopt (Core/LpopcIpopt.h:33-82).  Nothing here
computes; host arrays are numpy float64, device arrays torch CUDA tensors (HBM handles)."""
import ctypes as C

import numpy as np

from . import _abi
from .engine import RPM_E_INVALID, RPM_OK, RpmError, _dp, _ip, lib


class _EngineView:
    """rpm_group_engine(g, rank): set-up calls and options of one member engine (borrowed handle, never destroyed here)."""

    def __init__(self, L, h):
        self._L, self._h = L, h

    def get_option(self, key):
        v = C.c_int()
        rc = self._L.rpm_get_option(self._h, key.encode(), C.byref(v))
        if rc != RPM_OK:
            raise RpmError(rc, self._L.rpm_last_error(self._h).decode())
        return v.value


class EngineGroup:
    """One interval-sharded engine per entry of `devices` (a device may be listed more than once).  The host-consumer
    methods take caller-owned numpy arrays that every device addresses directly: they are page-locked on first sight and must
    stay allocated until close() or release_arrays()."""

    def __init__(self, problem, devices, options=None, n_instances=1):
        self._L = lib(getattr(problem.GetOpimalProblemFuns(), "library", None))
        self._desc, self._keep = _abi.lower(problem, options, n_instances, 0, 0, 1)
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        rc = self._L.rpm_group_create(C.byref(self._desc), len(devices), devs, C.byref(h))
        if rc != RPM_OK:
            raise RpmError(rc, self._L.rpm_group_last_error(None).decode())
        self._h = h
        self.size = self._L.rpm_group_size(h)
        e0 = self._L.rpm_group_engine(h, 0)
        n, m, nj, nh, st = (C.c_int() for _ in range(5))
        self._check(self._L.rpm_get_nlp_info(e0, C.byref(n), C.byref(m), C.byref(nj), C.byref(nh), C.byref(st)))
        self.n, self.m, self.nnz_jac, self.nnz_h = n.value, m.value, nj.value, nh.value
        self.n_instances = n_instances

    def close(self):
        if getattr(self, "_h", None):
            self._L.rpm_group_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc):
        if rc != RPM_OK:
            raise RpmError(rc, self._L.rpm_group_last_error(self._h).decode())

    def engine(self, rank):
        h = self._L.rpm_group_engine(self._h, int(rank))
        if not h:
            raise RpmError(RPM_E_INVALID, "no such rank")
        return _EngineView(self._L, h)

    def device_init(self):
        self._check(self._L.rpm_group_device_init(self._h))

    def set_option(self, key, value):
        self._check(self._L.rpm_group_set_option(self._h, key.encode(), int(value)))

    def release_arrays(self):
        """Let go of every page-locked caller array (before freeing them)."""
        self.set_option("pin_host", 0)

    def _x(self, x):
        if not (isinstance(x, np.ndarray) and x.dtype == np.float64 and x.flags.c_contiguous and x.size == self.n * self.n_instances):
            raise RpmError(RPM_E_INVALID, "x must be a caller-owned contiguous float64 array of n * n_instances entries")
        return x

    # ---- host consumer -----------------------------------------------------------------------------------------------
    def eval_f(self, x, new_x=True):
        out = np.zeros(self.n_instances)
        self._check(self._L.rpm_group_eval_f(self._h, self.n, _dp(self._x(x)), int(new_x), _dp(out)))
        return float(out[0]) if self.n_instances == 1 else out

    def eval_grad_f(self, x, out, new_x=True):
        self._check(self._L.rpm_group_eval_grad_f(self._h, self.n, _dp(self._x(x)), int(new_x), _dp(out)))
        return out

    def eval_g(self, x, out, new_x=True):
        self._check(self._L.rpm_group_eval_g(self._h, self.n, _dp(self._x(x)), int(new_x), self.m, _dp(out)))
        return out

    def eval_jac_g_structure(self):
        i, j = np.zeros(self.nnz_jac, dtype=np.int32), np.zeros(self.nnz_jac, dtype=np.int32)
        self._check(self._L.rpm_group_eval_jac_g(self._h, self.n, None, 0, self.m, self.nnz_jac, _ip(i), _ip(j), None))
        return i, j

    def eval_jac_g(self, x, out, new_x=True):
        self._check(self._L.rpm_group_eval_jac_g(self._h, self.n, _dp(self._x(x)), int(new_x), self.m, self.nnz_jac, None, None, _dp(out)))
        return out

    def eval_pair(self, x, g_out, values_out):
        self._check(self._L.rpm_group_eval_pair(self._h, self.n, _dp(self._x(x)), self.m, _dp(g_out), self.nnz_jac, _dp(values_out)))
        return g_out, values_out

    # ---- device consumer ---------------------------------------------------------------------------------------------
    def eval_pair_dev(self, home, d_x, d_g, d_values):
        """x, g, values in the HBM of rank `home`'s device; the other ranks store into them over xGMI.  Blocking."""
        self._check(self._L.rpm_group_eval_pair_dev(self._h, int(home), C.c_void_p(d_x.data_ptr()), C.c_void_p(d_g.data_ptr()),
                                                    C.c_void_p(d_values.data_ptr())))

    def allgather_pair_dev(self, d_x, d_g, d_values):
        """Lists of per-rank tensors (each on its rank's device); afterwards every rank's g / values are complete.  Blocking."""
        arr = lambda ts: (C.c_void_p * self.size)(*[t.data_ptr() for t in ts])   # noqa: E731
        self._check(self._L.rpm_group_allgather_pair_dev(self._h, arr(d_x), arr(d_g), arr(d_values)))


class SweepGroup:
    """rpm_sweep_*: the batched device solver (BatchedIPM) over several GPUs from one process — the B instances of one
    transcription dealt to `devices` in contiguous shares, solved side by side (a host thread per share inside the library)."""

    def __init__(self, problem, devices, n_instances, options=None, **solver_options):
        self._L = lib(getattr(problem.GetOpimalProblemFuns(), "library", None))
        self._desc, self._keep = _abi.lower(problem, options, n_instances, 0, 0, 1)
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        rc = self._L.rpm_sweep_create(C.byref(self._desc), len(devices), devs, C.byref(h))
        if rc != RPM_OK:
            raise RpmError(rc, self._L.rpm_sweep_last_error(None).decode())
        self._h = h
        self.size = self._L.rpm_sweep_size(h)
        self.n_instances = int(n_instances)
        e0 = self._L.rpm_sweep_engine(h, 0)
        n, m, nj, nh, st = (C.c_int() for _ in range(5))
        self._check(self._L.rpm_get_nlp_info(e0, C.byref(n), C.byref(m), C.byref(nj), C.byref(nh), C.byref(st)))
        self.n, self.m = n.value, m.value
        for k, v in solver_options.items():
            self.set_option(k, v)

    def close(self):
        if getattr(self, "_h", None):
            self._L.rpm_sweep_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc):
        if rc != RPM_OK:
            raise RpmError(rc, self._L.rpm_sweep_last_error(self._h).decode())

    def shares(self):
        out = []
        for r in range(self.size):
            a, b = C.c_int(), C.c_int()
            self._check(self._L.rpm_sweep_share(self._h, r, C.byref(a), C.byref(b)))
            out.append((a.value, b.value))
        return out

    def set_option(self, key, value):
        if key == "mu_strategy" and isinstance(value, str):
            value = {"monotone": 0, "adaptive": 1}[value]
        self._check(self._L.rpm_sweep_set_option(self._h, key.encode(), float(value)))

    def set_bounds(self, instance, x_l, x_u):
        x_l, x_u = np.ascontiguousarray(x_l, dtype=np.float64), np.ascontiguousarray(x_u, dtype=np.float64)
        assert x_l.size == self.n and x_u.size == self.n
        self._check(self._L.rpm_sweep_set_bounds(self._h, int(instance), _dp(x_l), _dp(x_u)))

    def solve(self, x0):
        B = self.n_instances
        x = np.ascontiguousarray(x0, dtype=np.float64).reshape(B, self.n).copy()
        lam = np.zeros((B, max(self.m, 1)))
        obj, err = np.zeros(B), np.zeros(B)
        status, its = np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32)
        self._check(self._L.rpm_sweep_solve(self._h, _dp(x), _dp(lam), _dp(obj), _ip(status), _ip(its), _dp(err)))
        return {"x": x, "lambda": lam[:, :self.m], "obj": obj, "status": status, "iterations": its, "kkt_error": err}

    def stats(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self._check(self._L.rpm_sweep_get_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"iterations": a.value, "factorizations": b.value, "trial_points": c.value}

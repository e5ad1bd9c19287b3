"""Python host side above the C ABI (include/rpm_hip.h): loads lpopc_amd/csrc/librpm_hip.so and
exposes the engine with the reference's TNLP method names (Core/LpopcIpopt.h:33-82).

There is no fallback of any kind: if the shared library is missing this module raises at
import of `lib()`, and every evaluation fails loudly when no HIP device is usable.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import _abi
from .problem import LpopcException

_HERE = os.path.dirname(os.path.abspath(__file__))
# RPM_HIP_LIB lets profiling sessions load the -DRPM_DIAG ablation build; it is never a different backend
_SO = os.environ.get("RPM_HIP_LIB") or os.path.join(_HERE, "csrc", "librpm_hip.so")
_LIB = None

RPM_OK, RPM_E_INVALID, RPM_E_UNSUPPORTED, RPM_E_DEVICE, RPM_E_NONFINITE = 0, 1, 2, 3, 4

# every symbol include/rpm_hip.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "rpm_create", "rpm_destroy", "rpm_last_error", "rpm_device_init", "rpm_get_nlp_info",
    "rpm_get_bounds_info", "rpm_get_starting_point", "rpm_eval_f", "rpm_eval_grad_f", "rpm_eval_g",
    "rpm_eval_jac_g", "rpm_eval_pair", "rpm_eval_h", "rpm_finalize_solution", "rpm_get_solution", "rpm_eval_g_dev",
    "rpm_eval_jac_g_dev", "rpm_eval_pair_dev", "rpm_eval_f_dev", "rpm_eval_grad_f_dev", "rpm_eval_h_dev",
    "rpm_synchronize", "rpm_set_option", "rpm_get_option", "rpm_set_instance_constants", "rpm_get_phase_sizes", "rpm_get_phase_tables",
    "rpm_shard_segments", "rpm_shard_pack_dev", "rpm_shard_unpack_dev", "rpm_shard_slot_len", "rpm_shard_pack_all_dev", "rpm_shard_unpack_all_dev", "rpm_nlp2op_control", "rpm_final_result_save",
    "rpm_solution_error", "rpm_ph_refine_mesh", "rpm_ph_refine_from_error",
    "rpm_hpliu_create", "rpm_hpliu_destroy", "rpm_hpliu_last_error", "rpm_hpliu_refine",
    "rpm_ipm_create", "rpm_ipm_destroy", "rpm_ipm_last_error", "rpm_ipm_set_option", "rpm_ipm_set_bounds", "rpm_ipm_set_all_bounds", "rpm_ipm_get_info",
    "rpm_group_create", "rpm_group_destroy", "rpm_group_last_error", "rpm_group_size", "rpm_group_engine", "rpm_group_device_init",
    "rpm_group_set_option", "rpm_group_eval_f", "rpm_group_eval_grad_f", "rpm_group_eval_g", "rpm_group_eval_jac_g", "rpm_group_eval_pair",
    "rpm_group_eval_h", "rpm_group_eval_pair_dev", "rpm_group_allgather_pair_dev",
    "rpm_sweep_create", "rpm_sweep_destroy", "rpm_sweep_last_error", "rpm_sweep_size", "rpm_sweep_engine", "rpm_sweep_solver", "rpm_sweep_share",
    "rpm_sweep_set_option", "rpm_sweep_set_bounds", "rpm_sweep_solve", "rpm_sweep_get_stats",
    "rpm_ipm_get_stats", "rpm_ipm_get_subproblems", "rpm_ipm_get_trace", "rpm_ipm_get_restorations", "rpm_ipm_get_kernel_times", "rpm_ipm_solve", "rpm_ipm_solve_dev", "rpm_ipm_get_permutation", "rpm_ipm_debug_solve", "rpm_ipm_debug_solve_dense", "rpm_ipm_debug_slot",
]


def build(force=False):
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", csrc] + (["-B"] if force else []) + ["librpm_hip.so"]
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return _SO


_LIBS = {}


def lib(path=None):
    """The native library (ctypes handle with its prototypes set): the package's librpm_hip.so, or — `path` — a library
    built from a user's functor header (lpopc_amd.userproblem).  One handle per path, loaded once."""
    global _LIB
    so = os.path.abspath(path) if path else _SO
    if so in _LIBS:
        return _LIBS[so]
    if not os.path.exists(so):
        raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(the engine has no CPU fallback)" % so)
    # One HIP runtime per process: PyTorch ships its own libamdhip64.so.7.  Importing torch first makes the
    # loader resolve this library's libamdhip64.so.7 dependency to the copy torch already mapped; loading in
    # the other order leaves two runtimes in the process and torch then reports "No HIP GPUs are available".
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(so)
    dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p
    L.rpm_create.argtypes = [C.POINTER(_abi.rpm_problem_desc), C.POINTER(vp)]
    L.rpm_destroy.argtypes = [vp]
    L.rpm_destroy.restype = None
    L.rpm_last_error.argtypes = [vp]
    L.rpm_last_error.restype = C.c_char_p
    L.rpm_device_init.argtypes = [vp, C.c_int]
    L.rpm_get_nlp_info.argtypes = [vp, ip, ip, ip, ip, ip]
    L.rpm_get_bounds_info.argtypes = [vp, C.c_int, dp, dp, C.c_int, dp, dp]
    L.rpm_get_starting_point.argtypes = [vp, C.c_int, C.c_int, dp, C.c_int, dp, dp, C.c_int, C.c_int, dp]
    L.rpm_eval_f.argtypes = [vp, C.c_int, dp, C.c_int, dp]
    L.rpm_eval_grad_f.argtypes = [vp, C.c_int, dp, C.c_int, dp]
    L.rpm_eval_g.argtypes = [vp, C.c_int, dp, C.c_int, C.c_int, dp]
    L.rpm_eval_jac_g.argtypes = [vp, C.c_int, dp, C.c_int, C.c_int, C.c_int, ip, ip, dp]
    L.rpm_eval_pair.argtypes = [vp, C.c_int, dp, C.c_int, dp, C.c_int, dp]
    L.rpm_eval_h.argtypes = [vp, C.c_int, dp, C.c_int, C.c_double, C.c_int, dp, C.c_int, C.c_int, ip, ip, dp]
    L.rpm_finalize_solution.argtypes = [vp, C.c_int, C.c_int, dp, dp, dp, C.c_int, dp, dp, C.c_double]
    L.rpm_get_solution.argtypes = [vp, C.c_int, dp, C.c_int, dp, dp]
    L.rpm_eval_g_dev.argtypes = [vp, vp, vp, vp]
    L.rpm_eval_jac_g_dev.argtypes = [vp, vp, vp, vp]
    L.rpm_eval_pair_dev.argtypes = [vp, vp, vp, vp, vp]
    L.rpm_eval_f_dev.argtypes = [vp, vp, vp, vp]
    L.rpm_eval_grad_f_dev.argtypes = [vp, vp, vp, vp]
    L.rpm_eval_h_dev.argtypes = [vp, vp, C.c_double, vp, vp, vp]
    L.rpm_synchronize.argtypes = [vp]
    L.rpm_set_option.argtypes = [vp, C.c_char_p, C.c_int]
    L.rpm_set_instance_constants.argtypes = [vp, C.c_int, dp, C.c_int]
    L.rpm_get_option.argtypes = [vp, C.c_char_p, ip]
    L.rpm_get_phase_sizes.argtypes = [vp, C.c_int, ip, ip, ip]
    L.rpm_get_phase_tables.argtypes = [vp, C.c_int, dp, dp, ip, ip, dp, dp, ip, ip, dp]
    L.rpm_nlp2op_control.argtypes = [vp, C.c_int, dp, dp, dp, dp, dp, dp, dp, dp, dp, dp]
    L.rpm_final_result_save.argtypes = [vp, C.c_char_p]
    L.rpm_solution_error.argtypes = [vp, C.c_int, dp, dp, ip]
    L.rpm_ph_refine_mesh.argtypes = [vp, C.c_int, dp, C.c_double, C.c_int, C.c_int, C.c_int, dp, ip, ip, dp, ip]
    L.rpm_ph_refine_from_error.argtypes = [vp, C.c_int, dp, C.c_double, C.c_int, C.c_int, C.c_int, dp, ip, ip, dp, ip]
    L.rpm_hpliu_create.argtypes = [C.c_int, C.c_double, C.c_int, C.c_double, C.POINTER(vp)]
    L.rpm_hpliu_destroy.argtypes = [vp]
    L.rpm_hpliu_destroy.restype = None
    L.rpm_hpliu_last_error.argtypes = [vp]
    L.rpm_hpliu_last_error.restype = C.c_char_p
    L.rpm_hpliu_refine.argtypes = [vp, vp, dp, dp, C.c_int, dp, ip, ip, ip, ip, ip]
    L.rpm_ipm_create.argtypes = [vp, C.POINTER(vp)]
    L.rpm_ipm_destroy.argtypes = [vp]
    L.rpm_ipm_destroy.restype = None
    L.rpm_ipm_last_error.argtypes = [vp]
    L.rpm_ipm_last_error.restype = C.c_char_p
    L.rpm_ipm_set_option.argtypes = [vp, C.c_char_p, C.c_double]
    L.rpm_ipm_set_bounds.argtypes = [vp, C.c_int, dp, dp]
    L.rpm_ipm_set_all_bounds.argtypes = [vp, dp, dp]
    L.rpm_ipm_get_info.argtypes = [vp, ip, ip, ip, ip, C.POINTER(C.c_longlong), ip]
    L.rpm_ipm_get_stats.argtypes = [vp, ip, ip, ip]
    L.rpm_ipm_get_subproblems.argtypes = [vp, C.c_int, ip, ip]
    L.rpm_ipm_get_trace.argtypes = [vp, C.c_int, C.c_int, dp, ip]
    L.rpm_ipm_get_restorations.argtypes = [vp, ip]
    L.rpm_ipm_get_kernel_times.argtypes = [vp, dp, dp]
    L.rpm_ipm_solve.argtypes = [vp, dp, dp, dp, ip, ip, dp]
    L.rpm_ipm_solve_dev.argtypes = [vp, vp, vp, dp, ip, ip, dp, vp]
    L.rpm_ipm_get_permutation.argtypes = [vp, ip, C.c_int]
    L.rpm_ipm_debug_solve.argtypes = [vp, dp, dp, dp, ip, ip]
    L.rpm_ipm_debug_solve_dense.argtypes = [vp, dp, dp, dp, ip, ip]
    L.rpm_ipm_debug_slot.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_longlong)]
    L.rpm_shard_segments.argtypes = [vp, C.c_int, C.c_int, C.POINTER(_abi.rpm_segment), ip, ip]
    L.rpm_shard_pack_dev.argtypes = [vp, C.c_int, vp, vp, vp]
    L.rpm_shard_unpack_dev.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp]
    L.rpm_shard_slot_len.argtypes = [vp, C.POINTER(C.c_longlong)]
    L.rpm_shard_pack_all_dev.argtypes = [vp, vp, vp, vp, vp]
    L.rpm_shard_unpack_all_dev.argtypes = [vp, vp, vp, vp, C.c_int, vp]
    L.rpm_group_create.argtypes = [C.POINTER(_abi.rpm_problem_desc), C.c_int, ip, C.POINTER(vp)]
    L.rpm_group_destroy.argtypes = [vp]
    L.rpm_group_destroy.restype = None
    L.rpm_group_last_error.argtypes = [vp]
    L.rpm_group_last_error.restype = C.c_char_p
    L.rpm_group_size.argtypes = [vp]
    L.rpm_group_engine.argtypes = [vp, C.c_int]
    L.rpm_group_engine.restype = vp
    L.rpm_group_device_init.argtypes = [vp]
    L.rpm_group_set_option.argtypes = [vp, C.c_char_p, C.c_int]
    L.rpm_group_eval_f.argtypes = [vp, C.c_int, dp, C.c_int, dp]
    L.rpm_group_eval_grad_f.argtypes = [vp, C.c_int, dp, C.c_int, dp]
    L.rpm_group_eval_g.argtypes = [vp, C.c_int, dp, C.c_int, C.c_int, dp]
    L.rpm_group_eval_jac_g.argtypes = [vp, C.c_int, dp, C.c_int, C.c_int, C.c_int, ip, ip, dp]
    L.rpm_group_eval_pair.argtypes = [vp, C.c_int, dp, C.c_int, dp, C.c_int, dp]
    L.rpm_group_eval_h.argtypes = [vp, C.c_int, dp, C.c_int, C.c_double, C.c_int, dp, C.c_int, C.c_int, ip, ip, dp]
    L.rpm_group_eval_pair_dev.argtypes = [vp, C.c_int, vp, vp, vp]
    L.rpm_group_allgather_pair_dev.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.rpm_sweep_create.argtypes = [vp, C.c_int, ip, C.POINTER(vp)]
    L.rpm_sweep_destroy.argtypes = [vp]
    L.rpm_sweep_destroy.restype = None
    L.rpm_sweep_last_error.argtypes = [vp]
    L.rpm_sweep_last_error.restype = C.c_char_p
    L.rpm_sweep_size.argtypes = [vp]
    L.rpm_sweep_engine.argtypes = [vp, C.c_int]
    L.rpm_sweep_engine.restype = vp
    L.rpm_sweep_solver.argtypes = [vp, C.c_int]
    L.rpm_sweep_solver.restype = vp
    L.rpm_sweep_share.argtypes = [vp, C.c_int, ip, ip]
    L.rpm_sweep_set_option.argtypes = [vp, C.c_char_p, C.c_double]
    L.rpm_sweep_set_bounds.argtypes = [vp, C.c_int, dp, dp]
    L.rpm_sweep_solve.argtypes = [vp, dp, dp, dp, ip, ip, dp]
    L.rpm_sweep_get_stats.argtypes = [vp, ip, ip, ip]
    _LIBS[so] = L
    if so == _SO:
        _LIB = L
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


class RpmError(LpopcException):
    def __init__(self, code, msg):
        super().__init__("rpm error %d: %s" % (code, msg))
        self.code = code


class NLPEngine:
    """One NLP (one mesh) on one GPU.  Method names and argument order follow LpopcIpopt
    (Core/LpopcIpopt.h:33-82); host arrays are numpy float64, device arrays are torch CUDA
    tensors (used only as HBM handles)."""

    def __init__(self, problem, options=None, n_instances=1, shard_mode=0, shard_rank=0, shard_world=1,
                 tile_nodes=0, device=None, role_loop=None):
        self._L = lib(getattr(problem.GetOpimalProblemFuns(), "library", None))
        self._desc, self._keep = _abi.lower(problem, options, n_instances, shard_mode, shard_rank, shard_world)
        h = C.c_void_p()
        rc = self._L.rpm_create(C.byref(self._desc), C.byref(h))
        if rc != RPM_OK:
            raise RpmError(rc, self._L.rpm_last_error(None).decode())
        self._h = h
        if tile_nodes:
            self.set_option("tile_nodes", tile_nodes)
        if role_loop is not None:
            self.set_option("role_loop", role_loop)
        # "pin_host" stays at the C ABI's default 0: this wrapper hands numpy arrays of any lifetime to the library, which
        # copies them through its own page-locked staging buffers.  A caller that reuses long-lived `out=` arrays (as
        # Ipopt does) may set_option("pin_host", 1) — then rpm_hip.h's lifetime contract for those arrays applies.
        n, m, nj, nh, st = (C.c_int() for _ in range(5))
        self._check(self._L.rpm_get_nlp_info(h, C.byref(n), C.byref(m), C.byref(nj), C.byref(nh), C.byref(st)))
        self.n, self.m, self.nnz_jac, self.nnz_h, self.index_style = n.value, m.value, nj.value, nh.value, st.value
        self.n_instances = n_instances
        self.n_phases = problem.GetPhaseNum()
        if device is not None:
            self.device_init(device)

    def close(self):
        if getattr(self, "_h", None):
            self._L.rpm_destroy(self._h)          # also releases every page-locked registration
            self._h = None
        self._bufs = {}                           # only now: see _own

    def __del__(self):
        self.close()

    def _own(self, name, size):
        """A result array owned by this object for calls without `out=`.  With pin_host = 1 the library page-locks the
        host arrays it is handed and holds the registration until eviction or close() (rpm_hip.h: such arrays must stay
        mapped that long); a temporary numpy array would be freed while still registered.  So temporaries never reach the
        library: results are written into these buffers, which live until close(), and handed out as copies."""
        bufs = self.__dict__.setdefault("_bufs", {})
        b = bufs.get(name)
        if b is None or b.size != size:
            if b is not None:
                bufs.setdefault("_retired", []).append(b)    # still registered: keep it alive
            b = bufs[name] = np.zeros(size)
        return b

    def _check(self, rc):
        if rc != RPM_OK:
            raise RpmError(rc, self._L.rpm_last_error(self._h).decode())

    def last_error(self):
        """rpm_last_error: the reason of the last failed call, plus what the page-lock registry last refused this engine."""
        return self._L.rpm_last_error(self._h).decode()

    def device_init(self, device_id=0):
        self._check(self._L.rpm_device_init(self._h, int(device_id)))

    def set_option(self, key, value):
        self._check(self._L.rpm_set_option(self._h, key.encode(), int(value)))

    def get_option(self, key):
        v = C.c_int()
        self._check(self._L.rpm_get_option(self._h, key.encode(), C.byref(v)))
        return v.value

    def set_instance_constants(self, instance, consts):
        """Give one instance of a batched engine its own problem constants (parameter sweeps)."""
        c = np.ascontiguousarray(consts, dtype=np.float64)
        self._check(self._L.rpm_set_instance_constants(self._h, int(instance), _dp(c), int(c.size)))

    # ---- TNLP surface, host buffers --------------------------------------------------------
    def get_nlp_info(self):
        return self.n, self.m, self.nnz_jac, self.nnz_h, self.index_style

    def get_bounds_info(self):
        xl, xu, gl, gu = np.zeros(self.n), np.zeros(self.n), np.zeros(self.m), np.zeros(self.m)
        self._check(self._L.rpm_get_bounds_info(self._h, self.n, _dp(xl), _dp(xu), self.m, _dp(gl), _dp(gu)))
        return xl, xu, gl, gu

    def get_starting_point(self):
        x = np.zeros(self.n)
        self._check(self._L.rpm_get_starting_point(self._h, self.n, 1, _dp(x), 0, None, None, self.m, 0, None))
        return x

    def _x(self, x):
        x0 = x
        x = np.ascontiguousarray(x, dtype=np.float64).ravel()
        if x.size != self.n * self.n_instances:
            raise RpmError(RPM_E_INVALID, "x has %d entries, expected %d" % (x.size, self.n * self.n_instances))
        if not (isinstance(x0, np.ndarray) and np.shares_memory(x, x0)):     # a converted copy: a temporary (see _own)
            b = self._own("x", x.size)
            b[:] = x
            return b
        return x

    def eval_f(self, x, new_x=True):
        x = self._x(x)
        out = self._own("f", self.n_instances)
        self._check(self._L.rpm_eval_f(self._h, self.n, _dp(x), int(new_x), _dp(out)))
        return float(out[0]) if self.n_instances == 1 else out.copy()

    def eval_grad_f(self, x, new_x=True, out=None):
        x = self._x(x)
        res = self._own("grad", self.n * self.n_instances) if out is None else out
        self._check(self._L.rpm_eval_grad_f(self._h, self.n, _dp(x), int(new_x), _dp(res)))
        return res.copy() if out is None else out

    def eval_g(self, x, new_x=True, out=None):
        x = self._x(x)
        res = self._own("g", self.m * self.n_instances) if out is None else out
        self._check(self._L.rpm_eval_g(self._h, self.n, _dp(x), int(new_x), self.m, _dp(res)))
        return res.copy() if out is None else out

    def eval_jac_g_structure(self):
        i, j = np.zeros(self.nnz_jac, dtype=np.int32), np.zeros(self.nnz_jac, dtype=np.int32)
        self._check(self._L.rpm_eval_jac_g(self._h, self.n, None, 0, self.m, self.nnz_jac, _ip(i), _ip(j), None))
        return i, j

    def eval_jac_g(self, x, new_x=True, out=None):
        x = self._x(x)
        res = self._own("values", self.nnz_jac * self.n_instances) if out is None else out
        self._check(self._L.rpm_eval_jac_g(self._h, self.n, _dp(x), int(new_x), self.m, self.nnz_jac, None, None, _dp(res)))
        return res.copy() if out is None else out

    def eval_pair(self, x, g_out=None, values_out=None):
        """rpm_eval_pair: eval_g + eval_jac_g of one x in one call (host arrays) -> (g, values)."""
        x = self._x(x)
        g = self._own("g", self.m * self.n_instances) if g_out is None else g_out
        v = self._own("values", self.nnz_jac * self.n_instances) if values_out is None else values_out
        self._check(self._L.rpm_eval_pair(self._h, self.n, _dp(x), self.m, _dp(g), self.nnz_jac, _dp(v)))
        return (g.copy() if g_out is None else g_out), (v.copy() if values_out is None else values_out)

    def eval_h_structure(self):
        i, j = np.zeros(self.nnz_h, dtype=np.int32), np.zeros(self.nnz_h, dtype=np.int32)
        self._check(self._L.rpm_eval_h(self._h, self.n, None, 0, 1.0, self.m, None, 0, self.nnz_h, _ip(i), _ip(j), None))
        return i, j

    def eval_h(self, x, obj_factor, lam, new_x=True, new_lambda=True):
        x = self._x(x)
        lam = np.ascontiguousarray(lam, dtype=np.float64)
        out = np.zeros(self.nnz_h * self.n_instances)
        self._check(self._L.rpm_eval_h(self._h, self.n, _dp(x), int(new_x), float(obj_factor), self.m, _dp(lam),
                                       int(new_lambda), self.nnz_h, None, None, _dp(out)))
        return out

    def finalize_solution(self, status, x, lam, obj_value):
        x = self._x(x)
        lam = np.ascontiguousarray(lam, dtype=np.float64)
        self._check(self._L.rpm_finalize_solution(self._h, int(status), self.n, _dp(x), None, None, self.m, None,
                                                  _dp(lam), float(obj_value)))

    def get_solution(self):
        x, lam, obj = np.zeros(self.n), np.zeros(self.m), C.c_double()
        self._check(self._L.rpm_get_solution(self._h, self.n, _dp(x), self.m, _dp(lam), C.byref(obj)))
        return x, lam, obj.value

    # ---- device-resident variants (torch CUDA tensors as HBM handles) -------------------------
    @staticmethod
    def _ptr(t):
        return C.c_void_p(t.data_ptr()) if t is not None else None

    @staticmethod
    def _stream(stream):
        if stream is None:
            import torch
            return C.c_void_p(torch.cuda.current_stream().cuda_stream)
        return C.c_void_p(stream)

    def eval_g_dev(self, d_x, d_g, stream=None):
        self._check(self._L.rpm_eval_g_dev(self._h, self._ptr(d_x), self._ptr(d_g), self._stream(stream)))

    def eval_jac_g_dev(self, d_x, d_values, stream=None):
        self._check(self._L.rpm_eval_jac_g_dev(self._h, self._ptr(d_x), self._ptr(d_values), self._stream(stream)))

    def eval_pair_dev(self, d_x, d_g, d_values, stream=None):
        self._check(self._L.rpm_eval_pair_dev(self._h, self._ptr(d_x), self._ptr(d_g), self._ptr(d_values),
                                              self._stream(stream)))

    def eval_f_dev(self, d_x, d_obj, stream=None):
        self._check(self._L.rpm_eval_f_dev(self._h, self._ptr(d_x), self._ptr(d_obj), self._stream(stream)))

    def eval_grad_f_dev(self, d_x, d_grad, stream=None):
        self._check(self._L.rpm_eval_grad_f_dev(self._h, self._ptr(d_x), self._ptr(d_grad), self._stream(stream)))

    def eval_h_dev(self, d_x, obj_factor, d_lambda, d_values, stream=None):
        self._check(self._L.rpm_eval_h_dev(self._h, self._ptr(d_x), float(obj_factor), self._ptr(d_lambda),
                                           self._ptr(d_values), self._stream(stream)))

    def synchronize(self):
        self._check(self._L.rpm_synchronize(self._h))

    # ---- solution extraction (Nlp2OpConverter) ----------------------------------------
    def nlp2op_control(self, phase, x=None, lam=None):
        d = self._desc.phases[phase]
        M = self.phase_tables(phase)["points"].size + 1
        out = dict(time=np.zeros(M), state=np.zeros(M * d.nx), control=np.zeros(M * max(d.nu, 1)),
                   costate=np.zeros(M * d.nx), pathmult=np.zeros(M * max(d.nc, 1)), hamiltonian=np.zeros(M))
        mc, lc = C.c_double(), C.c_double()
        xp = _dp(self._x(x)) if x is not None else None
        lp = _dp(np.ascontiguousarray(lam, dtype=np.float64)) if lam is not None else None
        self._check(self._L.rpm_nlp2op_control(self._h, phase, xp, lp, _dp(out["time"]), _dp(out["state"]),
                                               _dp(out["control"]), _dp(out["costate"]), _dp(out["pathmult"]),
                                               _dp(out["hamiltonian"]), C.cast(C.byref(mc), C.POINTER(C.c_double)),
                                               C.cast(C.byref(lc), C.POINTER(C.c_double))))
        out["control"] = out["control"][:M * d.nu]
        out["pathmult"] = out["pathmult"][:M * d.nc]
        out["mayer_cost"], out["lagrange_cost"] = mc.value, lc.value
        return out

    def final_result_save(self, directory):
        self._check(self._L.rpm_final_result_save(self._h, str(directory).encode()))

    # ---- mesh-error estimate and ph refinement ---------------------------------------
    def solution_error(self, phase, x=None):
        """SolutionErrorChecker::CheckSolutionDiffError -> relative_error, (N + K + 1) x nx."""
        rows = C.c_int()
        self._check(self._L.rpm_solution_error(self._h, phase, None, None, C.byref(rows)))
        nx = self._desc.phases[phase].nx
        rel = np.zeros(rows.value * nx)
        xp = _dp(self._x(x)) if x is not None else None
        self._check(self._L.rpm_solution_error(self._h, phase, xp, _dp(rel), C.byref(rows)))
        return rel.reshape((rows.value, nx), order="F")

    def _refine(self, fn, phase, arg, tol, nmin, nmax):
        K = self._desc.phases[phase].n_intervals
        nk, done, emax = C.c_int(), C.c_int(), np.zeros(K)
        self._check(fn(self._h, phase, arg, float(tol), int(nmin), int(nmax), 0, None, None, C.byref(nk), _dp(emax),
                       C.byref(done)))
        mesh, nodes = np.zeros(nk.value + 1), np.zeros(nk.value, dtype=np.int32)
        self._check(fn(self._h, phase, arg, float(tol), int(nmin), int(nmax), nk.value, _dp(mesh), _ip(nodes),
                       C.byref(nk), None, None))
        return bool(done.value), mesh, nodes, emax

    def ph_refine_mesh(self, phase, tol, nmin, nmax, x=None):
        """PhMeshRefineAlg::RefineMesh for one phase -> (no_more_refine, mesh_points, nodes_per_interval, emax)."""
        xp = _dp(self._x(x)) if x is not None else None
        return self._refine(self._L.rpm_ph_refine_mesh, phase, xp, tol, nmin, nmax)

    def ph_refine_from_error(self, phase, rel_err, tol, nmin, nmax):
        rel = np.asfortranarray(rel_err, dtype=np.float64).ravel(order="F").copy()
        return self._refine(self._L.rpm_ph_refine_from_error, phase, _dp(rel), tol, nmin, nmax)

    # ---- tables and sharding ---------------------------------------------------------------
    def phase_tables(self, phase):
        N, dn, on = C.c_int(), C.c_int(), C.c_int()
        self._check(self._L.rpm_get_phase_sizes(self._h, phase, C.byref(N), C.byref(dn), C.byref(on)))
        N, dn, on = N.value, dn.value, on.value
        t = dict(points=np.zeros(N), weights=np.zeros(N), d_rows=np.zeros(dn, dtype=np.int32),
                 d_cols=np.zeros(dn, dtype=np.int32), d_vals=np.zeros(dn), diag_vals=np.zeros(N),
                 doff_rows=np.zeros(on, dtype=np.int32), doff_cols=np.zeros(on, dtype=np.int32),
                 doff_vals=np.zeros(on))
        self._check(self._L.rpm_get_phase_tables(self._h, phase, _dp(t["points"]), _dp(t["weights"]),
                                                 _ip(t["d_rows"]), _ip(t["d_cols"]), _dp(t["d_vals"]),
                                                 _dp(t["diag_vals"]), _ip(t["doff_rows"]), _ip(t["doff_cols"]),
                                                 _dp(t["doff_vals"])))
        return t

    def shard_segments(self, which, rank):
        ns, pl = C.c_int(0), C.c_int(0)
        self._check(self._L.rpm_shard_segments(self._h, which, rank, None, C.byref(ns), C.byref(pl)))
        segs = (_abi.rpm_segment * max(ns.value, 1))()
        self._check(self._L.rpm_shard_segments(self._h, which, rank, segs, C.byref(ns), C.byref(pl)))
        return [(segs[i].off, segs[i].len, segs[i].pos) for i in range(ns.value)], pl.value

    def shard_pack_dev(self, which, d_full, d_packed, stream=None):
        self._check(self._L.rpm_shard_pack_dev(self._h, which, self._ptr(d_full), self._ptr(d_packed),
                                               self._stream(stream)))

    def shard_unpack_dev(self, which, d_gathered, stride, d_full, stream=None):
        self._check(self._L.rpm_shard_unpack_dev(self._h, which, self._ptr(d_gathered), int(stride),
                                                 self._ptr(d_full), self._stream(stream)))


    def shard_slot_len(self):
        v = C.c_longlong()
        self._check(self._L.rpm_shard_slot_len(self._h, C.byref(v)))
        return v.value

    def shard_pack_all_dev(self, d_g, d_values, d_slot, stream=None):
        self._check(self._L.rpm_shard_pack_all_dev(self._h, self._ptr(d_g), self._ptr(d_values), self._ptr(d_slot),
                                                   self._stream(stream)))

    def shard_unpack_all_dev(self, d_gathered, d_g, d_values, skip_own=True, stream=None):
        self._check(self._L.rpm_shard_unpack_all_dev(self._h, self._ptr(d_gathered), self._ptr(d_g), self._ptr(d_values),
                                                     1 if skip_own else 0, self._stream(stream)))


class HpLiuRefiner:
    """LiuHpMeshRefineAlg behind rpm_hpliu_* (Core/LpLiuHpMeshRefineAlg.cpp): one object per problem, it keeps the
    reference's mesh / solution histories across meshes."""

    def __init__(self, n_phases, tol, nmax, ratio_r):
        self._args = (int(n_phases), float(tol), int(nmax), float(ratio_r))
        self._L = None          # the native object is created by the library of the first engine it refines
        self._h = None
        self.P = int(n_phases)

    def _ensure(self, engine):
        if self._h is not None:
            if engine._L is not self._L:
                raise RpmError(RPM_E_INVALID, "HpLiuRefiner: engines of one refinement history must come from one native library")
            return
        self._L = engine._L
        self._h = C.c_void_p()
        rc = self._L.rpm_hpliu_create(*self._args, C.byref(self._h))
        if rc != RPM_OK:
            self._h = None
            raise RpmError(rc, "rpm_hpliu_create: invalid arguments")

    def close(self):
        if getattr(self, "_h", None):
            self._L.rpm_hpliu_destroy(self._h)
            self._h = None

    __del__ = close

    def refine(self, engine, x=None, rel_err=None, capacity=8192):
        """-> (no_more_refine, [(mesh_points, nodes_per_interval) per phase]).  rel_err: list of per-phase relative-error
        matrices to decide from (host only); default: estimate on the device."""
        self._ensure(engine)
        xp = _dp(engine._x(x)) if x is not None else None
        rp = None
        if rel_err is not None:
            flat = np.concatenate([np.asfortranarray(r, dtype=np.float64).ravel(order="F") for r in rel_err])
            rp = _dp(flat)
        mesh, nodes = np.zeros(capacity), np.zeros(capacity, dtype=np.int32)
        moff, noff, nk = (np.zeros(self.P, dtype=np.int32) for _ in range(3))
        done = C.c_int()
        rc = self._L.rpm_hpliu_refine(self._h, engine._h, xp, rp, capacity, _dp(mesh), _ip(nodes), _ip(moff), _ip(noff),
                                      _ip(nk), C.byref(done))
        if rc != RPM_OK:
            raise RpmError(rc, self._L.rpm_last_error(engine._h).decode())
        return bool(done.value), [(mesh[moff[p]:moff[p] + nk[p] + 1].copy(), nodes[noff[p]:noff[p] + nk[p]].copy())
                                  for p in range(self.P)]



class BatchedIPM:
    """rpm_ipm_* : the NLP solve (the reference's NLPSolver::SolveNlp -> Ipopt, Core/LpNLPSolver.cpp:13-53) for all of an
    engine's instances at once, iterates and KKT factors resident on the device.  The engine's hessian-approximation option
    decides what stands for the Hessian: "exact" lpopc's finite-difference Hessian (rpm_eval_h), "limited-memory" (lpopc's
    default) Ipopt's limited-memory BFGS."""

    STATUS = {0: "converged", 1: "converged to the acceptable level", 2: "iteration limit", 3: "line search and restoration phase failed",
              4: "inertia correction failed", 5: "NaN/Inf"}

    def __init__(self, engine, **options):
        self._L = engine._L
        self._e = engine
        self._h = C.c_void_p()
        rc = self._L.rpm_ipm_create(engine._h, C.byref(self._h))
        if rc != RPM_OK:
            raise RpmError(rc, self._L.rpm_last_error(engine._h).decode())
        for k, v in options.items():
            self.set_option(k, v)

    def close(self):
        if getattr(self, "_h", None):
            self._L.rpm_ipm_destroy(self._h)
            self._h = None

    __del__ = close

    def _chk(self, rc):
        if rc != RPM_OK:
            raise RpmError(rc, self._L.rpm_ipm_last_error(self._h).decode())

    def set_option(self, key, value):
        if key == "mu_strategy" and isinstance(value, str):       # Ipopt's spelling, as the restatement takes it
            value = {"monotone": 0, "adaptive": 1}[value]
        self._chk(self._L.rpm_ipm_set_option(self._h, key.encode(), float(value)))

    def set_bounds(self, instance, x_l, x_u):
        x_l, x_u = np.ascontiguousarray(x_l, dtype=np.float64), np.ascontiguousarray(x_u, dtype=np.float64)
        assert x_l.size == self._e.n and x_u.size == self._e.n
        self._chk(self._L.rpm_ipm_set_bounds(self._h, int(instance), _dp(x_l), _dp(x_u)))

    def set_all_bounds(self, x_l, x_u):
        """x_l, x_u: (n_instances, n)."""
        x_l, x_u = np.ascontiguousarray(x_l, dtype=np.float64), np.ascontiguousarray(x_u, dtype=np.float64)
        assert x_l.shape == x_u.shape == (self._e.n_instances, self._e.n)
        self._chk(self._L.rpm_ipm_set_all_bounds(self._h, _dp(x_l), _dp(x_u)))

    def info(self):
        a = [C.c_int() for _ in range(4)]
        st, ns = C.c_longlong(), C.c_int()
        self._chk(self._L.rpm_ipm_get_info(self._h, C.byref(a[0]), C.byref(a[1]), C.byref(a[2]), C.byref(a[3]), C.byref(st), C.byref(ns)))
        return {"kkt_order": a[0].value, "band_order": a[1].value, "half_bandwidth": a[2].value, "border": a[3].value,
                "storage_doubles": st.value, "n_slacks": ns.value}

    def subproblems(self):
        """Geometry of the factorisation's sub-problems: rows of (order, banded part, border, half bandwidth, column stride)."""
        n = C.c_int()
        self._chk(self._L.rpm_ipm_get_subproblems(self._h, 0, None, C.byref(n)))
        g = np.zeros((n.value, 5), dtype=np.int32)
        self._chk(self._L.rpm_ipm_get_subproblems(self._h, n.value, _ip(g), C.byref(n)))
        return g

    def factor_flops(self):
        """Floating-point operations of one LDL^T of one instance (multiply-adds counted as 2), from the sub-problems."""
        total = 0.0
        for nt, nb_, nbd, b, _ in self.subproblems():
            j = np.arange(nb_)
            r = np.minimum(b, nb_ - 1 - j) + nbd          # rows below the pivot of band column j
            total += float(np.sum(r * (r + 1.0))) + nbd ** 3 / 3.0
        return total

    def stats(self):
        a = [C.c_int() for _ in range(3)]
        self._chk(self._L.rpm_ipm_get_stats(self._h, C.byref(a[0]), C.byref(a[1]), C.byref(a[2])))
        return {"iterations": a[0].value, "factorizations": a[1].value, "trial_points": a[2].value}

    def trace(self, instance, capacity=4096):
        """Accepted steps of the last solve (option trace > 0): rows of f, theta, mu, alpha, alpha_z, delta_w, E_0, backtracks."""
        rec = np.zeros((capacity, 8))
        n = C.c_int()
        self._chk(self._L.rpm_ipm_get_trace(self._h, int(instance), capacity, _dp(rec), C.byref(n)))
        return rec[:n.value].copy()

    def kernel_times(self):
        """Device milliseconds of the last solve inside the factorisation / the substitution kernels (HIP events)."""
        f, s = C.c_double(), C.c_double()
        self._chk(self._L.rpm_ipm_get_kernel_times(self._h, C.byref(f), C.byref(s)))
        return {"factor_ms": f.value, "substitution_ms": s.value}

    def restorations(self):
        out = np.zeros(self._e.n_instances, dtype=np.int32)
        self._chk(self._L.rpm_ipm_get_restorations(self._h, _ip(out)))
        return out

    def permutation(self):
        nt = self.info()["kkt_order"]
        pos = np.zeros(nt, dtype=np.int32)
        self._chk(self._L.rpm_ipm_get_permutation(self._h, _ip(pos), nt))
        return pos

    def debug_solve(self, k_storage, rhs):
        B, nt = self._e.n_instances, self.info()["kkt_order"]
        k = np.ascontiguousarray(k_storage, dtype=np.float64)
        r = np.ascontiguousarray(rhs, dtype=np.float64)
        assert k.size == B * self.info()["storage_doubles"] and r.size == B * nt
        sol = np.zeros((B, nt))
        npos, nneg = np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32)
        self._chk(self._L.rpm_ipm_debug_solve(self._h, _dp(k), _dp(r), _dp(sol), _ip(npos), _ip(nneg)))
        return sol, npos, nneg

    def debug_solve_dense(self, k_dense, rhs):
        """Factor + solve dense symmetric matrices (B, Nt, Nt) in unknown order; -> (solutions, n_pos, n_neg)."""
        B, nt = self._e.n_instances, self.info()["kkt_order"]
        k = np.ascontiguousarray(k_dense, dtype=np.float64)
        r = np.ascontiguousarray(rhs, dtype=np.float64)
        assert k.shape == (B, nt, nt) and r.shape == (B, nt)
        sol = np.zeros((B, nt))
        npos, nneg = np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32)
        self._chk(self._L.rpm_ipm_debug_solve_dense(self._h, _dp(k), _dp(r), _dp(sol), _ip(npos), _ip(nneg)))
        return sol, npos, nneg

    def slot(self, ua, uc):
        off = C.c_longlong()
        self._chk(self._L.rpm_ipm_debug_slot(self._h, int(ua), int(uc), C.byref(off)))
        return off.value

    def solve(self, x0):
        """x0: (n_instances, n) starting points -> dict(x, lambda, obj, status, iterations, kkt_error)."""
        B, n, m = self._e.n_instances, self._e.n, self._e.m
        x = np.array(x0, dtype=np.float64, order="C").reshape(B, n)
        lam = np.zeros((B, max(m, 1)))[:, :m].copy()
        obj, err = np.zeros(B), np.zeros(B)
        status, iters = np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32)
        self._chk(self._L.rpm_ipm_solve(self._h, _dp(x), _dp(lam), _dp(obj), _ip(status), _ip(iters), _dp(err)))
        return {"x": x, "lambda": lam, "obj": obj, "status": status, "iterations": iters, "kkt_error": err}

    def solve_dev(self, d_x, d_lambda=None, stream=None):
        """d_x: torch CUDA tensor (n_instances, n), overwritten with the solutions; d_lambda: optional (n_instances, m).
        The solver waits for what is queued on `stream` (default: torch's current stream) before it touches the arrays."""
        B = self._e.n_instances
        obj, err = np.zeros(B), np.zeros(B)
        status, iters = np.zeros(B, dtype=np.int32), np.zeros(B, dtype=np.int32)
        lp = C.c_void_p(d_lambda.data_ptr()) if d_lambda is not None else None
        self._chk(self._L.rpm_ipm_solve_dev(self._h, C.c_void_p(d_x.data_ptr()), lp, _dp(obj), _ip(status), _ip(iters), _dp(err),
                                            NLPEngine._stream(stream)))
        return {"obj": obj, "status": status, "iterations": iters, "kkt_error": err}

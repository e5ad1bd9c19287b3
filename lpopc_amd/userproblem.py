"""Your own problem without touching the library's sources — the stand-in for subclassing lpopc's FunctionWrapper
(Core/LpFunctionWrapper.h:50-69: MayerCost / LagrangeCost / DaeFunction / EventFunction / LinkFunction and their Deriv*
twins).  The callbacks have to run inside the GPU kernels, so they are a header with ONE struct,

    namespace rpm { struct UserProblem { NX, NU, NC, NE_MAX, NLINK_MAX, NCONST, HAS_ANALYTIC; dae(...); event(...);
                                         link(...); mayer(...); lagrange(...); [*_jac_col / *_grad_col] }; }

(interface and worked examples: lpopc_amd/csrc/problems/problems.hpp, examples/user_problem_vanderpol.hpp), and
`build()` compiles the engine's kernels around it for gfx950 with the same hipcc the package was built with, into a
library of its own, cached by the hash of the header and of the engine's sources:

    lib = userproblem.build("my_problem.hpp")
    fun = ProblemFunctor(RPM_PROBLEM_USER, consts, library=lib)          # where the reference takes shared_ptr<FunctionWrapper>
    prob = OptimalProblem(n_phases, n_links, fun) ... NLPEngine(prob)

A C++ host links or dlopens the produced library instead of librpm_hip.so.  Nothing here computes."""
import hashlib
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
RPM_PROBLEM_USER = 100


def _sources():
    names = []
    for root, _, files in os.walk(_CSRC):
        if os.path.basename(root).startswith("build") or "user_libs" in root:
            continue
        for f in files:
            if f.endswith((".cpp", ".hip", ".hpp", ".h")) or f == "Makefile":
                names.append(os.path.join(root, f))
    names.append(os.path.join(_HERE, "..", "include", "rpm_hip.h"))
    return sorted(names)


def build(header, cache_dir=None, only_user=True, jobs=8, verbose=False):
    """Compile the engine around the functor `rpm::UserProblem` defined in `header` -> path of the shared library.
    only_user: leave the built-in problems out of that library (a 6x shorter build; they stay available in librpm_hip.so).

    The cache key is the hash of the header's text and of the engine's sources; files the header #includes are NOT tracked
    (keep a functor in one header, or pass a fresh cache_dir after editing an include).  Concurrent callers (two ranks under
    torchrun) are serialised by a lock on the cache directory; the library is linked under a temporary name and renamed into
    place, so nobody ever loads a half-written file."""
    import fcntl
    header = os.path.abspath(header)
    if not os.path.exists(header):
        raise FileNotFoundError(header)
    if any(c in header for c in " \t\n'\"\\$`"):
        raise ValueError("the header path is spliced into a make command line: no whitespace, quotes, backslashes or $ (%r)" % header)
    h = hashlib.sha256()
    h.update(open(header, "rb").read())
    h.update(b"only_user" if only_user else b"all")
    for s in _sources():
        h.update(open(s, "rb").read())
    tag = h.hexdigest()[:16]
    cache_dir = os.path.abspath(cache_dir or os.path.join(_CSRC, "user_libs"))
    os.makedirs(cache_dir, exist_ok=True)
    so = os.path.join(cache_dir, "librpm_hip_user_%s.so" % tag)
    if os.path.exists(so):
        return so
    with open(os.path.join(cache_dir, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if os.path.exists(so):                       # another process built it while this one waited
                return so
            tmp = os.path.join(cache_dir, "librpm_hip_user_%s.%d.tmp.so" % (tag, os.getpid()))
            objdir = os.path.join(cache_dir, "build_%s" % tag)
            extra = '-DRPM_USER_PROBLEM_HEADER=\\"%s\\"' % header + (" -DRPM_ONLY_USER_PROBLEM" if only_user else "")
            cmd = ["make", "-C", _CSRC, "-j%d" % jobs, "LIB=%s" % tmp, "OBJDIR=%s" % objdir, "EXTRA=%s" % extra, tmp]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0 or not os.path.exists(tmp):
                raise RuntimeError("building the library for %s failed:\n%s\n%s" % (header, r.stdout[-2000:], r.stderr[-6000:]))
            os.replace(tmp, so)
            if verbose:
                print(r.stdout[-1000:])
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return so

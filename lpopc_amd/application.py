"""Host-side mirror of lpopc's application shell, for end-to-end runs of the GPU path.

Reference: `LpopcApplication` (Core/LpLpopcApplication.hpp:28-36: SetOptimalControlProblem / Options /
SolveOptimalProblem) and the outer loop of `LpopcAlgorithm::Optimization` (Core/LpLpopcAlgorithm.cpp:20-43):

    SetFirstMesh; GetSizes; GetBounds; GetGuess; SolveNlp; Nlp2OpControl;
    while (!RefineMesh()) { UpdateGrid; GetSizes; GetBounds; GetGuess; SolveNlp; Nlp2OpControl; }
    FinalResultSave

`GetSizes/GetBounds/GetGuess` are what `rpm_create` does per mesh; `Nlp2OpControl`, the error estimate and the
refinement decision are the post-solve entry points of the C ABI.  The NLP solver itself is NOT part of the reference's
sources (it hands an `Ipopt::TNLP` to Ipopt 3.12.3, Core/LpNLPSolver.cpp:13-53) and Ipopt is not in this image:
`DeviceIPMSolver` (hessian-approximation=exact) runs a restatement of Ipopt's published algorithm on the device
(rpm_ipm_*, row f-2); `ScipyNLPSolver` drives the same callbacks with scipy's trust-constr instead.  It is a stand-in for small problems (tests,
demos), not a replacement for Ipopt; in particular its constraint multipliers are not of Ipopt's quality, so costates and
Hamiltonian extracted from them are indicative only.
"""
import numpy as np

from .engine import NLPEngine
from .mesh import MeshRefiner, install_guess
from .problem import LpopcException, Options

console_not_print, console_print = 0, 1


class ScipyNLPSolver:
    """Stand-in for NLPSolver::SolveNlp (Core/LpNLPSolver.cpp:13-53): min f(x) s.t. g_l <= g(x) <= g_u, x_l <= x <= x_u
    through the TNLP callbacks (eval_f, eval_grad_f, eval_g, eval_jac_g + structure), quasi-Newton Hessian like the
    reference's default hessian-approximation=limited-memory."""

    def __init__(self, tol=1e-6, maxiter=1500):
        self.tol, self.maxiter = float(tol), int(maxiter)

    def SolveNlp(self, nlp):
        from scipy.optimize import BFGS, Bounds, NonlinearConstraint, minimize
        from scipy.sparse import coo_matrix

        xl, xu, gl, gu = nlp.get_bounds_info()
        i, j = nlp.eval_jac_g_structure()
        n, m = nlp.n, nlp.m

        def jac(x):
            return coo_matrix((nlp.eval_jac_g(x), (i, j)), shape=(m, n)).tocsr()

        try:   # the KKT systems here are small: a BLAS thread pool only gets in its own way (100x on a many-core host)
            from threadpoolctl import threadpool_limits
            limit = threadpool_limits(limits=1)
        except ImportError:
            limit = None
        try:
            res = minimize(lambda x: float(np.ravel(nlp.eval_f(x))[0]), np.clip(nlp.get_starting_point(), xl, xu),
                           jac=nlp.eval_grad_f, hess=BFGS(), bounds=Bounds(xl, xu),
                           constraints=[NonlinearConstraint(nlp.eval_g, gl, gu, jac=jac)], method="trust-constr",
                           options={"maxiter": self.maxiter, "gtol": self.tol * 1e-2, "xtol": 1e-12, "verbose": 0})
        finally:
            if limit is not None:
                limit.restore_original_limits()
        lam = np.asarray(res.v[0], dtype=np.float64) if len(res.v) else np.zeros(m)
        nlp.finalize_solution(int(res.status), res.x, lam, float(res.fun))
        self.last = res
        # trust-constr creeps through its last digits on some problems; a feasible point at the iteration limit is
        # still reported (the caller sees res.status in self.last)
        return res.status in (1, 2) or res.constr_violation <= 1e-5


class DeviceIPMSolver:
    """NLPSolver::SolveNlp (Core/LpNLPSolver.cpp:13-53) on the device: rpm_ipm_* restates the interior-point algorithm of
    the Ipopt the reference calls (see include/rpm_hip.h, row f-2), with "tol" = the Ipopt-tol option exactly as the
    reference passes it, and the engine's hessian-approximation (exact: eval_h; limited-memory: BFGS pairs).  Its multipliers are the NLP's (lambda of the
    primal-dual system), so costates and the Hamiltonian extracted from them are meaningful."""

    def __init__(self, tol=1e-6, maxiter=3000, retry_bound_relax=(1e-7, 1e-6), **solver_options):
        self.tol, self.maxiter = float(tol), int(maxiter)
        # Degenerate bounds (Delta-III: masses pinned to their bounds by the dynamics) make the multipliers non-unique; now and
        # then a path ends with a slack collapsed against its 1e-8-relaxed bound and the line search stalls (status 3; DESIGN.md
        # f-2).  The same solve with the bounds moved out a little further takes another path: an explicit, recorded retry here,
        # not something the C ABI does behind the caller's back.
        self.retry_bound_relax = tuple(retry_bound_relax)
        self.solver_options = solver_options

    def SolveNlp(self, nlp):
        from .engine import BatchedIPM
        x0 = nlp.get_starting_point()
        self.attempts = []
        for relax in (None,) + self.retry_bound_relax:
            opts = dict(self.solver_options)
            if relax is not None:
                opts["bound_relax_factor"] = relax
            ipm = BatchedIPM(nlp, tol=self.tol, max_iter=self.maxiter, **opts)
            try:
                r = ipm.solve(x0)
                self.last = dict(r, stats=ipm.stats(), info=ipm.info())
            finally:
                ipm.close()
            self.attempts.append((relax, int(r["status"][0]), int(r["iterations"][0])))
            if int(r["status"][0]) in (0, 1):
                break
        nlp.finalize_solution(int(r["status"][0]), r["x"][0], r["lambda"][0], float(r["obj"][0]))
        return int(r["status"][0]) in (0, 1)      # Solve_Succeeded / Solved_To_Acceptable_Level


class LpopcApplication:
    def __init__(self, if_console_print=console_print):
        self.print_ = if_console_print
        self.optionlist_ = Options()
        self.optpro_ = None
        self.result = None          # per phase: the arrays of Nlp2OpControl on the final mesh
        self.objective = None
        self.meshrefiner_ = None

    def SetOptimalControlProblem(self, user_optimal_control_problem):
        self.optpro_ = user_optimal_control_problem

    def Options(self):
        return self.optionlist_

    def _say(self, msg):
        if self.print_:
            print(msg)

    def CheckAnalyticDerive(self, device=0):
        """analytic-derive-check=yes (LpANDeriveChecker::CheckeAnlyticlDerive, Core/LpANDeriveChecker.cpp:13-571; wired in
        Core/LpLpopcAlgorithm.cpp:179-184): at the guess, the user's analytic derivatives against forward differences with
        perturbation analytic-derive-check-tol, entries differing by more than that tolerance are reported.  The reference
        compares the raw derivative matrices of every user function; here the comparison is made on what those matrices
        become — the NLP Jacobian values and the objective gradient — so a (dt/2) factor sits on the dynamics entries.
        Returns the list of (kind, index, analytic, finite difference) that differ; prints them like the reference warns."""
        from .problem import Options
        tol = self.optionlist_.GetNumericValue("analytic-derive-check-tol")
        fd = Options()
        fd.SetNumericValue("finite-difference-tol", tol)
        an = Options()
        an.SetStringValue("first-derive", "analytic")
        bad = []
        ea, ef = NLPEngine(self.optpro_, an, device=device), NLPEngine(self.optpro_, fd, device=device)
        try:
            x = ea.get_starting_point()
            for kind, a, f in (("jacobian value", ea.eval_jac_g(x), ef.eval_jac_g(x)),
                               ("objective gradient", ea.eval_grad_f(x), ef.eval_grad_f(x))):
                for k in np.nonzero(np.abs(a - f) > tol)[0]:
                    bad.append((kind, int(k), float(a[k]), float(f[k])))
                    self._say("%s %d: \tuser=%16f,\t finite difference =%16f,\t error=%16f" % (kind, k, a[k], f[k], abs(a[k] - f[k])))
        finally:
            ea.close()
            ef.close()
        return bad

    def SolveOptimalProblem(self, nlp_solver=None, device=0, result_dir=None):
        if self.optpro_ is None:
            raise LpopcException("No optimal control problem has been set")
        if nlp_solver is not None:
            solver = nlp_solver
        else:
            # the whole solve stays on the device, with lpopc's exact (finite-difference) Hessian or — its default,
            # Core/LpNLPWrapper.hpp:71 — Ipopt's limited-memory BFGS (csrc/rpm_ipm_lbfgs.hip).  ScipyNLPSolver remains for
            # callers that ask for it (nlp_solver=ScipyNLPSolver(...)).
            solver = DeviceIPMSolver(self.optionlist_.GetNumericValue("Ipopt-tol"))
        self.last_solver = solver
        self.meshrefiner_ = MeshRefiner(self.optionlist_)
        if (self.optionlist_.GetStringValue("first-derive") == "analytic"
                and self.optionlist_.GetStringValue("analytic-derive-check") == "yes"):
            self._say("Checking user defined analytic derivatives against finite difference")
            self.derive_check = self.CheckAnalyticDerive(device)
        while True:
            eng = NLPEngine(self.optpro_, self.optionlist_, device=device)     # GetSizes, GetBounds, GetGuess
            try:
                ok = solver.SolveNlp(eng)
                self.objective = eng.get_solution()[2]
                self._say("grid %d: n=%d m=%d objective %.9g%s" % (self.meshrefiner_.CurrentGrid(), eng.n, eng.m, self.objective,
                                                                  "" if ok else " (NLP solver did not converge)"))
                install_guess(eng, self.optpro_)                                   # Nlp2OpControl
                self.result = [eng.nlp2op_control(i) for i in range(self.optpro_.GetPhaseNum())]
                done = self.meshrefiner_.RefineMesh(eng, self.optpro_)             # error estimate + new mesh
                if done and result_dir is not None:
                    eng.final_result_save(result_dir)                             # FinalResultSave
            finally:
                eng.close()
            if done:
                self._say("Optimal Problem Solved!Lpopc Exited!")
                return True

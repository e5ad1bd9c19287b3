/*
 * rpm_hip.h — C ABI of the MI355X-native Radau-pseudospectral NLP-callback engine.
 *
 * This is the drop-in boundary for lpopc's per-iteration hot path.  Every entry
 * point replaces one interface of the reference (paths relative to
 * /root/reference/Lpopc/src, cited per function below).  Plain C types only:
 * `int` is Ipopt::Index, `double` is Ipopt::Number (Core/LpopcIpopt.h:33-82).
 *
 * Conventions
 *   - every function returns 0 on success and a non-zero RPM_E_* code on failure;
 *     no C++ exception ever crosses this boundary (the reference throws
 *     LP_THROW_EXCEPTION, Common/LpException.hpp:78).  The message of the last
 *     failure is available from rpm_last_error().
 *   - the set-up half (rpm_create, rpm_get_nlp_info, rpm_get_bounds_info,
 *     rpm_get_starting_point, structure pass of rpm_eval_jac_g / rpm_eval_h,
 *     rpm_get_phase_tables, rpm_shard_*) is host-only and needs no GPU.
 *   - the evaluation half runs hand-written HIP kernels on gfx950.  There is NO
 *     CPU fallback: without a usable device these calls fail with RPM_E_DEVICE.
 *   - an engine may hold `n_instances` structurally identical OCP instances
 *     (the batched MPC sweep).  All vectors are then instance-major:
 *     x[inst*n + i], g[inst*m + i], values[inst*nnz + k].  n/m/nnz reported by
 *     rpm_get_nlp_info are always per instance.
 */
#ifndef RPM_HIP_H_
#define RPM_HIP_H_

#ifdef __cplusplus
extern "C" {
#endif

#define RPM_ABI_VERSION 2

/* ---- error codes -------------------------------------------------------- */
enum {
  RPM_OK = 0,
  RPM_E_INVALID = 1,  /* bad argument / inconsistent description (LpSizeChecker, LpBoundsChecker,
                         LpGuessChecker, LpMeshRefiner exceptions in the reference) */
  RPM_E_UNSUPPORTED = 2,
  RPM_E_DEVICE = 3,   /* HIP error or no device */
  RPM_E_NONFINITE = 4 /* a result contains NaN/Inf (the reference always returns true) */
};

/* ---- problem functor ids ------------------------------------------------- */
/* The reference's user callbacks are host C++ virtuals on Armadillo matrices
 * (Core/LpFunctionWrapper.h:50-69); they cannot run on a GPU.  Here the five
 * callbacks (Mayer, Lagrange, Dae, Event, Link) are pointwise device functors
 * compiled into the library and selected by id. */
enum {
  RPM_PROBLEM_LAUNCH = 1,          /* Delta-III ascent, example/launch/Launch.cpp:620-765 */
  RPM_PROBLEM_HYPERSENSITIVE = 2,  /* example/hypersensitive/HyperSensitive.cpp:74-167 */
  RPM_PROBLEM_BRYSON_DENHAM = 3,   /* example/bryson-denham/BrysonDenham.cpp:100-167 */
  RPM_PROBLEM_BRACHISTOCHRONE = 4, /* authored here (BASELINE config 1) */
  RPM_PROBLEM_MIN_TIME_CLIMB = 5,  /* authored here (BASELINE config 2) */
  RPM_PROBLEM_QUADROTOR = 6,       /* authored here (BASELINE config 5) */
  RPM_PROBLEM_PARAM_SLED = 7,      /* authored here: minimum-time sled with one static parameter (nq = 1), known optimum 2.5 */
  RPM_PROBLEM_PARAM_OSC = 8,       /* authored here: two linked phases, two static parameters each (nq = 2), path + events + links on them */
  RPM_PROBLEM_USER = 100           /* the functor `rpm::UserProblem` of a user's header, in a library built from that header
                                      (lpopc_amd/userproblem.py, INTEGRATION.md "your own problem"): the stand-in for
                                      subclassing FunctionWrapper (Core/LpFunctionWrapper.h:50-69) */
};

/* first-derive option, Core/LpOptDerive.hpp:29-32 */
enum { RPM_DERIVE_FINITE_DIFFERENCE = 0, RPM_DERIVE_ANALYTIC = 1 };
/* hessian-approximation option, Core/LpNLPWrapper.hpp:71-72 */
enum { RPM_HESSIAN_LIMITED_MEMORY = 0, RPM_HESSIAN_EXACT = 1 };
/* how work is split when shard_world > 1 */
enum { RPM_SHARD_NONE = 0, RPM_SHARD_INTERVALS = 1 };

/* ---- problem description (what Phase/Linkage/OptimalProblem setters collect,
 *      Core/LpOptimalProblem.hpp:30-326) ------------------------------------- */
typedef struct rpm_phase_desc {
  int nx, nu, nq, nc, ne;          /* Phase(idx,nx,nu,nq,nc,ne)  :33-37 */
  int n_intervals;                 /* SetMeshPoints/SetNodesPerInterval :174-187 */
  const double* mesh_points;       /* n_intervals+1, must run -1 .. +1 (Core/LpMeshRefiner.cpp:40) */
  const int* nodes_per_interval;   /* n_intervals, each >= 2 */
  double t0_min, tf_min;           /* SetTimeMin(t0,tf) :49 */
  double t0_max, tf_max;           /* SetTimeMax(t0,tf) :57 */
  const double* state_min;         /* nx*3: {state0,state,statef} per state, SetStateMin :65 */
  const double* state_max;         /* nx*3 */
  const double* control_min;       /* nu */
  const double* control_max;
  const double* parameter_min;     /* nq */
  const double* parameter_max;
  const double* path_min;          /* nc */
  const double* path_max;
  const double* event_min;         /* ne */
  const double* event_max;
  int has_duration;                /* SetDuration :126 */
  double duration_min, duration_max;
  int n_guess;                     /* number of guess knots (>=2), SetTimeGuess :132 */
  const double* time_guess;        /* n_guess */
  const double* state_guess;       /* nx*n_guess, knot-fastest per state (SetStateGuess(i,v)) */
  const double* control_guess;     /* nu*n_guess */
  const double* parameter_guess;   /* nq */
} rpm_phase_desc;

typedef struct rpm_link_desc {
  int left_phase, right_phase;     /* 1-based, Linkage(ipair,left,right) :245 */
  int n_links;
  const double* link_min;          /* n_links */
  const double* link_max;
} rpm_link_desc;

typedef struct rpm_problem_desc {
  int abi_version;                 /* RPM_ABI_VERSION */
  int problem_id;                  /* RPM_PROBLEM_* */
  int n_phases;
  const rpm_phase_desc* phases;
  int n_links;
  const rpm_link_desc* links;
  int n_consts;                    /* problem constants handed to the functor (the reference keeps
                                      them in globals, e.g. CONSTANTS in example/launch/Launch.cpp:47-74) */
  const double* consts;
  double fd_tol;                   /* finite-difference-tol, default 1e-6 */
  int first_derive;                /* RPM_DERIVE_* */
  int hessian_approximation;       /* RPM_HESSIAN_* */
  int n_instances;                 /* >=1; structurally identical OCPs evaluated per call */
  int shard_mode;                  /* RPM_SHARD_* */
  int shard_rank, shard_world;     /* this engine computes only the tiles it owns */
} rpm_problem_desc;

typedef struct rpm_engine rpm_engine;

/* ---- lifecycle ------------------------------------------------------------
 * rpm_create does what LpopcAlgorithm::GetSizes/GetBounds/GetGuess do once per mesh
 * (Core/LpLpopcAlgorithm.cpp:143-148): size/bounds/guess checks, NLP layout
 * (Core/LpBoundsChecker.cpp:13-348), collocation tables (Core/RPMGenerator.cpp:43-181),
 * Jacobian structure (Core/LpNLPWrapper.cpp:1106-1578).  Host only. */
int rpm_create(const rpm_problem_desc* desc, rpm_engine** out);
void rpm_destroy(rpm_engine* e);
/* message of the last failed call on this engine (or of the last failed rpm_create when e==NULL) */
const char* rpm_last_error(const rpm_engine* e);
/* bind the engine to HIP device `device_id`, allocate device tables/buffers, create its stream.
 * Called implicitly (device 0) by the first evaluation if omitted. */
int rpm_device_init(rpm_engine* e, int device_id);

/* ---- Ipopt::TNLP surface (Core/LpopcIpopt.h:33-82, Core/LpopcIpopt.cpp) ------- */
/* LpopcIpopt::get_nlp_info, LpopcIpopt.cpp:11-24.  index_style: 0 = C_STYLE. */
int rpm_get_nlp_info(rpm_engine* e, int* n, int* m, int* nnz_jac_g, int* nnz_h_lag, int* index_style);
/* LpopcIpopt::get_bounds_info, LpopcIpopt.cpp:26-82 */
int rpm_get_bounds_info(rpm_engine* e, int n, double* x_l, double* x_u, int m, double* g_l, double* g_u);
/* LpopcIpopt::get_starting_point, LpopcIpopt.cpp:84-104 (requires init_x=1, init_z=0, init_lambda=0) */
int rpm_get_starting_point(rpm_engine* e, int n, int init_x, double* x, int init_z, double* z_L,
                           double* z_U, int m, int init_lambda, double* lambda);
/* LpopcIpopt::eval_f, LpopcIpopt.cpp:106-116 -> NLPWrapper::GetObjFun, LpNLPWrapper.cpp:863 */
int rpm_eval_f(rpm_engine* e, int n, const double* x, int new_x, double* obj_value);
/* LpopcIpopt::eval_grad_f, LpopcIpopt.cpp:118-133 -> GetObjGrad, LpNLPWrapper.cpp:940 */
int rpm_eval_grad_f(rpm_engine* e, int n, const double* x, int new_x, double* grad_f);
/* LpopcIpopt::eval_g, LpopcIpopt.cpp:135-150 -> GetAllCons, LpNLPWrapper.cpp:34 */
int rpm_eval_g(rpm_engine* e, int n, const double* x, int new_x, int m, double* g);
/* LpopcIpopt::eval_jac_g, LpopcIpopt.cpp:152-181: values==NULL -> structure pass (iRow/jCol,
 * 0-based), else values pass -> GetConsJacbi, LpNLPWrapper.cpp:230 */
int rpm_eval_jac_g(rpm_engine* e, int n, const double* x, int new_x, int m, int nele_jac, int* iRow,
                   int* jCol, double* values);
/* Both constraint callbacks in one call (host pointers): what Ipopt's back-to-back eval_g(x, new_x = true) +
 * eval_jac_g(x, new_x = false) pair (LpopcIpopt.cpp:135-181) asks for, for callers that can take both results at once
 * (an SQP / own interior-point loop, the MPC sweep's host driver): one launch writes g straight into the caller's
 * page-locked array and the Jacobian values into HBM, a second queue operation delivers `values`, ONE synchronisation.
 * Same results as the two calls, same options ("pin_host", "zero_copy", "const_once", "delta_values"). */
int rpm_eval_pair(rpm_engine* e, int n, const double* x, int m, double* g, int nele_jac, double* values);
/* LpopcIpopt::eval_h, LpopcIpopt.cpp:183-218 -> LpHessianCalculator::GetHessian, LpHessian.cpp:878 */
int rpm_eval_h(rpm_engine* e, int n, const double* x, int new_x, double obj_factor, int m,
               const double* lambda, int new_lambda, int nele_hess, int* iRow, int* jCol,
               double* values);
/* LpopcIpopt::finalize_solution, LpopcIpopt.cpp:220-246: keeps x, lambda, obj for rpm_get_solution */
int rpm_finalize_solution(rpm_engine* e, int status, int n, const double* x, const double* z_L,
                          const double* z_U, int m, const double* g, const double* lambda,
                          double obj_value);
int rpm_get_solution(rpm_engine* e, int n, double* x, int m, double* lambda, double* obj_value);

/* ---- solution extraction (the step after the NLP solve) -----------------------
 * Nlp2OpConverter::Nlp2OpControl, Core/Nlp2OPConverter.cpp:13-196, for one phase: time, states, controls (with the
 * spline-extrapolated value at tau=+1 appended), costates -W^-1 lambda (and -D(:,N)' lambda at the end point), path
 * multipliers, Hamiltonian, Mayer and Lagrange cost.  Outputs are host arrays with N+1 rows, column-major; any may
 * be NULL.  x / lambda are host arrays; NULL = the solution stored by rpm_finalize_solution.  The time/state/
 * control arrays are also what the reference installs as the next mesh's guess (:149-193). */
int rpm_nlp2op_control(rpm_engine* e, int phase, const double* x, const double* lambda, double* time, double* state,
                       double* control, double* costate, double* pathmult, double* hamiltonian, double* mayer_cost,
                       double* lagrange_cost);
/* Nlp2OpConverter::FinalResultSave, Core/Nlp2OPConverter.cpp:198-223: writes time<k>, state<k>, control<k>, parameter<k>,
 * costate<k>, Hamiltonian<k> (Armadillo raw_ascii: one row per line) for every phase k into `dir`. */
int rpm_final_result_save(rpm_engine* e, const char* dir);

/* ---- mesh-error estimate and ph mesh refinement (after extraction) ------------
 * SolutionErrorChecker::CheckSolutionDiffError, Core/LpSolutionError.cpp:112-169, for one phase: the solution is
 * interpolated onto a mesh with one more LGR point per interval (:46-108), the dynamics are integrated there with
 * inv(D(:,1:)) (Core/RPMGenerator.cpp:85), and rel_err = |integrated - interpolated| / (1 + max of the state's column).
 * rel_err is a host array, (N + K + 1) rows x nx, column-major (K = mesh intervals); *rows receives the row count
 * (rel_err may be NULL to query it).  x is a host array; NULL = the solution stored by rpm_finalize_solution. */
int rpm_solution_error(rpm_engine* e, int phase, const double* x, double* rel_err, int* rows);
/* PhMeshRefineAlg::RefineMesh + ModifySegment, Core/LpPhMeshRefineAlg.cpp:12-100, for one phase: an interval whose
 * largest relative error is <= tol is kept; otherwise it gets Pq = int(log(emax/tol)/log(n)) more nodes, or, when
 * that exceeds nmax, is split into max(ceil((n+Pq)/nmin), 2) intervals of nmin nodes.  Outputs (any may be NULL):
 * new_mesh_points (new_n_intervals + 1), new_nodes_per_interval (new_n_intervals; both need `capacity` >= that
 * count, query it first with NULL arrays), interval_error (K, the per-interval maxima), no_more_refine (1 when
 * every interval met tol: the reference's NoMoreRefine).  The caller installs the new mesh with a new rpm_create,
 * as the reference re-runs GetSizes/GetBounds/GetGuess per mesh (Algorithm/LpLpopcAlgorithm.cpp:147-169). */
int rpm_ph_refine_mesh(rpm_engine* e, int phase, const double* x, double tol, int nmin, int nmax, int capacity,
                       double* new_mesh_points, int* new_nodes_per_interval, int* new_n_intervals,
                       double* interval_error, int* no_more_refine);

/* The refinement decision alone, from a relative_error matrix the caller already holds (host only, no device work). */
int rpm_ph_refine_from_error(rpm_engine* e, int phase, const double* rel_err, double tol, int nmin, int nmax,
                             int capacity, double* new_mesh_points, int* new_nodes_per_interval,
                             int* new_n_intervals, double* interval_error, int* no_more_refine);

/* hp-Liu refinement: LiuHpMeshRefineAlg::RefineMesh, Core/LpLiuHpMeshRefineAlg.cpp:12-260 (with Reducing_N :438-481,
 * Increasing_N :379-436, Dividing_mesh :321-377, CanWeIncreaseN :606-681; Merging_mesh's verdict is unused by the
 * reference, equal-N satisfied neighbours always merge).  The object keeps the reference's histories (meshes with their
 * per-interval errors, previous solution, previous mesh points) across meshes: create it once per problem
 * (tol = desired-relative-error, nmax = Nmax, ratio_r = R, Core/LpMeshRefiner.h:54-61), call rpm_hpliu_refine after every
 * solve with the engine built on the mesh the previous call returned.  Outputs for all phases: phase p's new mesh
 * points at new_mesh_points[mesh_off[p] .. + new_n_intervals[p]], node counts at new_nodes_per_interval[nodes_off[p]
 * ..]; `capacity` entries each.  rel_err: NULL = estimate on the device (rpm_solution_error), else the caller's matrices
 * phase after phase (host only).  Fails with RPM_E_INVALID + message where the reference would throw or hit an
 * undefined cast (see csrc/rpm_hpliu.cpp). */
typedef struct rpm_hpliu rpm_hpliu;
int rpm_hpliu_create(int n_phases, double tol, int nmax, double ratio_r, rpm_hpliu** out);
void rpm_hpliu_destroy(rpm_hpliu* h);
const char* rpm_hpliu_last_error(const rpm_hpliu* h);
int rpm_hpliu_refine(rpm_hpliu* h, rpm_engine* e, const double* x, const double* rel_err, int capacity,
                     double* new_mesh_points, int* new_nodes_per_interval, int* mesh_off, int* nodes_off,
                     int* new_n_intervals, int* no_more_refine);

/* ---- row f-2: the NLP solve itself, batched and device-resident ---------------------------------------------------
 * The reference hands the TNLP to Ipopt 3.12.3 (NLPSolver::SolveNlp, Core/LpNLPSolver.cpp:13-53: "tol" from the
 * Ipopt-tol option, hessian_approximation from the option list; Ipopt is a third-party dependency that is not in the
 * reference tree).  rpm_ipm restates Ipopt's published algorithm (Waechter & Biegler 2006: primal-dual barrier,
 * fraction-to-the-boundary rule, filter line search with second-order correction, inertia correction, l1 restoration
 * phase, monotone or adaptive barrier update; no NLP scaling, no watchdog, no quality-function oracle) for the engine's
 * n_instances independent NLPs at once — the MPC sweep — with iterates, multipliers, the KKT matrices and their LDL^T
 * factors resident in HBM; per iteration only a few counters cross PCIe.  The engine's hessian_approximation decides what
 * stands for the Hessian of the Lagrangian: RPM_HESSIAN_EXACT = lpopc's finite-difference Hessian (rpm_eval_h);
 * RPM_HESSIAN_LIMITED_MEMORY — lpopc's default, Core/LpNLPWrapper.hpp:71 — = Ipopt's limited-memory BFGS (history 6, scaling
 * s'y / s's, its skipping rule; csrc/rpm_ipm_lbfgs.hip): the KKT matrix of a diagonal Hessian is factored and the low-rank
 * part enters every solve through the Sherman-Morrison-Woodbury formula (12 more substitutions per iteration); also the only
 * mode for problems with static parameters (nq > 0).  Set the engine's "instance_align" before rpm_ipm_create.
 *   rpm_ipm_set_option: "tol" (1e-8), "max_iter" (3000), "mu_init" (0.1), "bound_push", "bound_frac" (1e-2),
 *                       "delta_c" (1e-9, constraint regularisation that makes the pivot-free LDL^T well defined; keep it
 *                       well below bound_relax_factor),
 *                       "max_line_search" (40), "trace" (0; keep the first N accepted steps of every instance),
 *                       "restoration" (1), "restoration_max_iter" (300), "restoration_penalty" (1000, Ipopt's rho),
 *                       "acceptable_tol" (1e-6), "acceptable_iter" (15), "bound_relax_factor" (1e-8, as Ipopt: finite bounds
 *                       of free unknowns move out by this * max(1, |bound|), so a solution may sit that far outside them),
 *                       "max_soc" (4, second-order correction steps per iteration; 0 = off),
 *                       "mu_strategy" (1, default: adaptive = what lpopc asks Ipopt for, Core/LpNLPSolver.cpp:28, with the
 *                       LOQO oracle and the kkt-error globalisation; 0: the monotone Fiacco-McCormick rule,
 *                       three batched iterations fewer on the quadrotor sweep),
 *                       "sigma_cap" (0 = off; experimental clamp on z/s in the KKT matrix),
 *                       "init_ls_multipliers" (0; 1 = least-squares multipliers at the first iterate, Ipopt's default start),
 *                       "level1_dense" (1 where it applies: the interval blocks of the nested dissection are factored out of
 *                       registers, kkt_factor_dense_kernel, when each has at most 21 block rows of 16 (17 resident, the first
 *                       4 block columns through the storage); 0 = the left-looking kernel; 1 on a layout it does not fit:
 *                       RPM_E_UNSUPPORTED),
 *                       "upper_dense" (1 where it applies: the last level of the nested dissection — and the groups of
 *                       separators of a three-level layout when their band is at least half as wide as long — on that kernel
 *                       too, the last level with its border x border corner eliminated there as well (panels of the corner's
 *                       block columns by substitution); 2 = the corner by the left-looking kernel's unblocked elimination
 *                       (bit for bit what 0 gives); 0 = the left-looking kernel for them; results agree to rounding),
 *                       "fused_fill" (1 where level1_dense runs: that kernel assembles its interval block from the Jacobian,
 *                       Hessian and diagonal terms itself and carries the level-1 forward substitution of the iteration's
 *                       right-hand side along, the fill kernel leaves level-1 storage alone; 0 = separate kernels; same
 *                       results bit for bit; 1 where level 1 does not run on that kernel: RPM_E_UNSUPPORTED),
 *                       "nlp_scaling" (0; 1 = Ipopt's gradient-based NLP scaling, ITS default: objective and constraint rows
 *                       scaled so that no gradient entry at the starting point exceeds "nlp_scaling_max_gradient" (100); the
 *                       multipliers and the objective come back unscaled; DESIGN.md f-2 on why it is off here),
 *                       "ic_hot_start" (0; 1 = an iteration whose predecessor needed delta_w > 0 starts the inertia correction
 *                       at kappa_w^- * delta_w_last instead of 0 while that is >= "ic_hot_min" (1e-10): not Ipopt's rule, an
 *                       experiment)
 *   rpm_ipm_set_bounds: variable bounds of one instance (default: the engine's); the fixed/free pattern is shared
 *   rpm_ipm_solve[_dev]: x (n_instances x n, in: starting points, out: solutions; host resp. device pointer),
 *                       lambda (n_instances x m, may be NULL); per instance on the host, any may be NULL: objective,
 *                       status (0 converged, 1 converged to Ipopt's acceptable level, 2 iteration limit, 3 line search failed where Ipopt would enter
 *                       restoration, 4 inertia correction failed, 5 NaN/Inf), iteration count, scaled KKT error
 *   rpm_ipm_get_info:   order of the KKT system, of its banded part, half bandwidth, border size, doubles of storage per
 *                       instance, number of slack variables (one per inequality row) */
typedef struct rpm_ipm rpm_ipm;
int rpm_ipm_create(rpm_engine* e, rpm_ipm** out);
void rpm_ipm_destroy(rpm_ipm* s);
const char* rpm_ipm_last_error(const rpm_ipm* s);
int rpm_ipm_set_option(rpm_ipm* s, const char* key, double value);
int rpm_ipm_set_bounds(rpm_ipm* s, int instance, const double* x_l, const double* x_u);
int rpm_ipm_set_all_bounds(rpm_ipm* s, const double* x_l, const double* x_u);   /* n_instances x n each */
int rpm_ipm_get_info(rpm_ipm* s, int* kkt_order, int* band_order, int* half_bandwidth, int* border,
                     long long* storage_doubles, int* n_slacks);
int rpm_ipm_get_stats(rpm_ipm* s, int* iterations, int* factorizations, int* trial_points);
/* the factorisation's sub-problems: 5 ints each (order, banded part, border, half bandwidth, doubles per stored column); one
 * entry for the band + border layout, the interval blocks followed by the separator system with nested dissection */
int rpm_ipm_get_subproblems(rpm_ipm* s, int capacity, int* geom, int* n_sub);
/* records of the last solve when option "trace" > 0: 8 doubles per accepted step — f, theta = |c|_1, mu, alpha, alpha_z,
 * delta_w, E_0 at the step's start, backtracking steps */
int rpm_ipm_get_trace(rpm_ipm* s, int instance, int capacity, double* records, int* n_records);
/* restoration phases each instance went through in the last solve (option "restoration", default 1: when the line search
 * gives up at an infeasible point the instance switches, as Ipopt does, to  min rho |p + n|_1 + zeta/2 |D_R (v - v_R)|^2
 * s.t. c(v) - p + n = 0, p, n >= 0 and the bounds  — solved by the same interior-point kernels with p, n eliminated from the
 * Newton system — until the infeasibility is 0.9 of where it entered and the original filter accepts the point; least-squares
 * multipliers on return.  Trace records of restoration iterations carry -1 in the backtracking field.  DESIGN.md f-2) */
int rpm_ipm_get_restorations(rpm_ipm* s, int* per_instance);
/* device time of the last solve spent in the factorisation and in the substitution kernels (HIP events on the solver's
 * stream, summed over its iterations), for roofline figures */
int rpm_ipm_get_kernel_times(rpm_ipm* s, double* factor_ms, double* substitution_ms);
int rpm_ipm_solve(rpm_ipm* s, double* x, double* lambda, double* obj, int* status, int* iterations, double* kkt_error);
/* Ordering contract of the device-resident form: the interior-point loop runs on the engine's private stream.  Before its
 * first read of d_x (and first write of d_lambda) it waits for everything the caller has queued on `stream` (a hipStream_t,
 * NULL = the legacy default stream) up to this call; it returns only after its own stream has drained, so on return the
 * results in d_x / d_lambda are complete for the host and for work queued later on any stream. */
int rpm_ipm_solve_dev(rpm_ipm* s, double* d_x, double* d_lambda, double* obj, int* status, int* iterations,
                      double* kkt_error, void* stream);
/* test hooks: KKT position of every unknown ([0,n) variables, slacks, then the m multipliers); factor + solve the
 * caller's matrices given in the band + border storage (host pointers, n_instances of each) */
int rpm_ipm_get_permutation(rpm_ipm* s, int* pos, int capacity);
int rpm_ipm_debug_solve(rpm_ipm* s, const double* k_storage, const double* rhs, double* sol, int* n_pos, int* n_neg);
/* layout-independent forms (band + border, or nested dissection: engine option "ipm_nested" = 1 before rpm_ipm_create — every
 * mesh interval is eliminated by a workgroup of its own up to the states at its first node, the Schur complements add up into
 * a block-tridiagonal separator system + border; same LDL^T without pivoting, same inertia from the signs of D): dense
 * symmetric matrices B x Nt x Nt and vectors in unknown order ([0,n) variables, slacks, multipliers); rpm_ipm_debug_slot
 * tells whether the layout can hold entry (ua, uc) */
int rpm_ipm_debug_solve_dense(rpm_ipm* s, const double* k_dense, const double* rhs, double* sol, int* n_pos, int* n_neg);
int rpm_ipm_debug_slot(rpm_ipm* s, int ua, int uc, long long* offset);

/* ---- device-resident variants (inputs/outputs already in HBM; used by benches, by the
 *      MPC sweep and by any device-side solver).  Pointers are device pointers on the
 *      engine's device; `stream` is a hipStream_t with HIP's own meaning (NULL = the legacy
 *      default stream).  Calls are asynchronous on that stream; the host-pointer entry points
 *      above use a private stream of the engine instead. ------------------------------- */
int rpm_eval_g_dev(rpm_engine* e, const double* d_x, double* d_g, void* stream);
int rpm_eval_jac_g_dev(rpm_engine* e, const double* d_x, double* d_values, void* stream);
/* fused pair: one launch produces g and the Jacobian values of the same x */
int rpm_eval_pair_dev(rpm_engine* e, const double* d_x, double* d_g, double* d_values, void* stream);
int rpm_eval_f_dev(rpm_engine* e, const double* d_x, double* d_obj, void* stream);
int rpm_eval_grad_f_dev(rpm_engine* e, const double* d_x, double* d_grad_f, void* stream);
int rpm_eval_h_dev(rpm_engine* e, const double* d_x, double obj_factor, const double* d_lambda,
                   double* d_values, void* stream);
/* block until everything queued on the engine's stream has finished */
int rpm_synchronize(rpm_engine* e);

/* ---- engine options ------------------------------------------------------------
 * key                values
 * "fuse_pair"        1 (default): rpm_eval_g(new_x=1) runs the fused pair kernel and the following
 *                    rpm_eval_jac_g(new_x=0) on the same x returns the cached values; 0: separate kernels
 * "dx_mode"          0: scalar ascending-column D.X (bit-identical to the reference's COO loop,
 *                    SparseMatrix/LpSparseMatrix.cpp:142-153); 1: v_mfma_f64_16x16x4 tiles
 * "tile_nodes"       16 | 32 | 64: collocation nodes per workgroup (0 = default 16)
 * "role_loop"        -1 (default): automatic, 0: never, 1: always — the throughput thread layout (64 nodes x 4 role
 *                    groups per workgroup, roles walked sequentially) chosen automatically for large grids
 * "pipeline"         -1 (default): automatic, 0: never, 1: whenever the mesh fits — with the role-looped layout, run the
 *                    persistent pipelined kernel (4 compute waves + 1 DMA wave per workgroup; inputs of the next tile
 *                    prefetched, constant block written by the DMA wave); automatic = every resident workgroup has
 *                    at least two tiles.  get-only "pipeline_active": 1 if the next launch uses it
 * "stage_roles"      -1 (default) | 0 | 1: in the pipelined kernel, problem functors that offer their dynamics in stages
 *                    (csrc/problems/problems.hpp `has_stage`: the launch vehicle, the quadrotor) are evaluated in full once per
 *                    node and per perturbation role only in what the perturbed variable enters — the same operations, the
 *                    same bits; -1: where the launch skips the constant block ("persistent_values"), which is bound by the
 *                    dynamics; with all stores the extra registers cost more than the arithmetic saves
 * "instance_align"   1 (default) or any power of two up to 65536 doubles: in the device-resident calls with n_instances > 1 the g /
 *                    values arrays of consecutive instances are rpm_get_option "stride_g" / "stride_values" doubles
 *                    apart (m, nnz_jac rounded up to this multiple) instead of packed back to back, so that every
 *                    instance starts on a 64/128-byte boundary like a separately allocated array; x stays packed; the
 *                    host-pointer entry points keep the packed layout
 * "const_once"       0 (default) | 1: host-pointer rpm_eval_jac_g downloads the linear and constant tail of `values`
 *                    (they never change, LpNLPWrapper.cpp:242, :715-718) only into a buffer it did not fill on the
 *                    previous call, afterwards just the NL prefix — for callers that hand the same array every
 *                    iteration and leave it alone in between (Ipopt's TNLPAdapter does); one instance per engine
 * "persistent_values" 0 (default) | 1: rpm_eval_jac_g_dev / rpm_eval_pair_dev remember the device `values` arrays they have
 *                    completely written and, called again with the same array, write only the entries that depend on x (and
 *                    the 2(P+L) linear ones): the constant Doffdiag block (Core/LpNLPWrapper.cpp:715-718; 54 % of the
 *                    metric problem's entries) is neither loaded nor stored again — SURVEY.md section 8(d)'s persistent-
 *                    buffer byte count B'.  Contract: the caller leaves that block of the array alone and does not free and
 *                    re-allocate the array in between; setting any option forgets every array (do that after re-allocating).
 *                    Bit-identical `values`.  The device solver (rpm_ipm_*) uses it for its own Jacobian array.
 * "delta_values"     0 (default) | 1: host-pointer rpm_eval_jac_g / rpm_eval_pair deliver `values` by difference: a kernel
 *                    compares the fresh values with a device-side mirror of what this engine last stored into the SAME
 *                    host array and stores only the 512-double runs in which a bit changed — the constant Doffdiag block, the
 *                    linear entries and every finite-difference block that does not depend on x (LpNLPWrapper.cpp:722-728 stores
 *                    them although `dependencies` is all ones, :291) cross PCIe once.  Contract as for "const_once": the caller
 *                    hands the same array and leaves it alone between calls (Ipopt's TNLPAdapter does); the engine checks 64
 *                    sampled entries of the array before every delivery and falls back to a full delivery when one differs or
 *                    the array is a different one.  Needs "pin_host" (the array is page-locked and mapped).  Results are
 *                    bit-identical to a full delivery.  get-only "delta_total_runs": runs this engine owns; "delta_sent_runs": runs
 *                    stored since the previous query of this option (blocking; for tests and reports).
 *                    Interval-sharded engines deliver only the runs they own (host-consumer multi-GPU mode).
 * "ipm_nested"       -1 (default: when the structure allows) | 0 | 1 (must), read by rpm_ipm_create: factor the KKT matrices by
 *                    nested dissection over the mesh intervals (one workgroup per interval and instance instead of one per
 *                    instance: the metric problem's single instance uses 256 CUs instead of one; 0 = one band + border matrix)
 * "ipm_nested_group" 0 (default: automatic) | positions per group: with nested dissection the separator system (block
 *                    tridiagonal along time) of a long mesh is cut once more into groups of this many of its positions, each
 *                    eliminated by a workgroup of its own (automatic: when that system is >= 512 long, groups of ~sqrt(length * bandwidth))
 * "ipm_local_border" 1 (default) | 0, read by rpm_ipm_create: with nested dissection an interval's block carries rows only for the
 *                    unknowns of the global border that its interior has entries with (its phase's t0 and tf, the phase's final
 *                    states for the last interval); 0: every interval carries the whole border (rows of zeros in L)
 * "zero_copy"        1 (default): the tile kernel reads x straight from page-locked host memory and stores g straight into it
 *                    (the caller's arrays with "pin_host", else the engine's staging buffers) — no copy-engine operations,
 *                    one launch + one synchronisation per rpm_eval_g; 0: through the engine's HBM buffers with copy-engine transfers
 * "check_finite"     1 (default): NaN/Inf in a result -> RPM_E_NONFINITE (checked on the device); 0: lpopc's behaviour
 * "pin_host"         0 (default) | 1.  0: the host-pointer entry points copy x into, and g / values / grad_f out of,
 *                    page-locked staging buffers the engine owns (CPU copies); the caller's memory is never handed to
 *                    the HIP runtime, so arrays of any lifetime may be passed (this replaces LpopcIpopt's own heap copy
 *                    of x and element-wise copy-out, Core/LpopcIpopt.cpp:135-181).  1 (opt-in; RpmTNLP asks for it on
 *                    behalf of Ipopt, whose TNLPAdapter hands the same arrays every iteration): arrays of >= 64 KB are
 *                    page-locked (hipHostRegister) the first time they are seen, the kernels read x from and store g into
 *                    them, copy engines and the delta delivery write `values` straight into them.  Registrations live in
 *                    ONE process-wide table shared by every engine of the process (librpm_pin.so): exactly the arrays'
 *                    bytes, never a byte twice, overlapping arrays as one range; an engine holds at most 8 arrays
 *                    (least recently used is let go first) and lets go of all of them in rpm_destroy or when the option
 *                    is set to 0; a range is unregistered when its last holder lets go.  A request the runtime refuses,
 *                    or that partly overlaps memory another engine holds, is served through the staging buffers instead and
 *                    is never silent: get-only "pin_register_failures", "pin_unregister_failures", "pin_overlap_refused"
 *                    (process-wide counts; also "pin_registered", "pin_unregistered", "pin_shared", "pin_merged",
 *                    "pin_evicted", "pin_live", and "pin_held" = this engine's) and the reason in rpm_last_error.
 *                    LIFETIME (option 1 only): a registered array must stay allocated until rpm_destroy, its eviction or
 *                    that release — set the option to 0 BEFORE freeing or unmapping such an array.
 */
/* Parameter sweeps (n_instances > 1): by default every instance shares the problem functor's constants
 * (rpm_problem_desc.consts — the reference keeps them in file-scope globals, example/launch/Launch.cpp:47-74).  This
 * gives instance `instance` its own copy (n = the functor's constant count), e.g. one tracking target per MPC problem;
 * all batched device-resident entry points and rpm_ipm_* then evaluate every instance with its own constants.  The
 * one-instance post-solve entry points keep using instance 0's. */
int rpm_set_instance_constants(rpm_engine* e, int instance, const double* consts, int n);
int rpm_set_option(rpm_engine* e, const char* key, int value);
int rpm_get_option(rpm_engine* e, const char* key, int* value);

/* ---- collocation tables of one phase (struct ps, Core/LpCalculateData.hpp:35-41; built like
 *      RPMGenerator::initialize, Core/RPMGenerator.cpp:43-105).  Any output may be NULL.
 *      D is returned as the reference's COO triplets in its own order. ---------------- */
int rpm_get_phase_sizes(rpm_engine* e, int phase, int* n_nodes, int* d_nnz, int* doff_nnz);
int rpm_get_phase_tables(rpm_engine* e, int phase, double* points, double* weights, int* d_rows,
                         int* d_cols, double* d_vals, double* diag_vals, int* doff_rows,
                         int* doff_cols, double* doff_vals);

/* ---- interval sharding helpers (shard_mode = RPM_SHARD_INTERVALS) --------------------
 * A rank's share of g / values is a list of contiguous runs.  Segment s of rank r is
 * dst[off .. off+len) of the full vector and sits at packed offset `pos` of that rank's
 * contiguous send buffer.  which: 0 = g, 1 = jacobian values.  Pass seg==NULL to query the
 * count.  Host only. */
typedef struct rpm_segment { int off, len, pos; } rpm_segment;
int rpm_shard_segments(rpm_engine* e, int which, int rank, rpm_segment* seg, int* n_seg,
                       int* packed_len);
/* pack this rank's runs of a full-size device vector into a contiguous device buffer / scatter a
 * gathered [world][max_packed_len] buffer back into TNLP order */
int rpm_shard_pack_dev(rpm_engine* e, int which, const double* d_full, double* d_packed, void* stream);
int rpm_shard_unpack_dev(rpm_engine* e, int which, const double* d_gathered, int stride,
                         double* d_full, void* stream);

/* ONE collective per step: the runs of g AND of the Jacobian values of ALL the engine's instances that this rank owns go
 * into one slot of rpm_shard_slot_len doubles (the same for every rank, whole 128-byte lines); the caller all-gathers the
 * [world][slot] buffer in place (RCCL; bench.py captures pack, all-gather and unpack in the step's hipGraph) and
 * rpm_shard_unpack_all_dev scatters it into TNLP order (skip_own = 1: this rank's own runs are in place already).  Instance
 * b's share of rank r's slot is [g runs | values runs] at offset b * (packed_len_g(r) + packed_len_values(r)); d_g / d_values
 * use the engine's instance strides ("instance_align").  Bit-identical to the single-GPU vectors (no reductions). */
int rpm_shard_slot_len(rpm_engine* e, long long* slot_doubles);
int rpm_shard_pack_all_dev(rpm_engine* e, const double* d_g, const double* d_values, double* d_slot, void* stream);
int rpm_shard_unpack_all_dev(rpm_engine* e, const double* d_gathered, double* d_g, double* d_values, int skip_own,
                             void* stream);

/* ---- one process, several GPUs: the mesh intervals of ONE NLP sharded over the devices of a node -------------------------
 * The caller this is for is lpopc's NLPSolver::SolveNlp (Core/LpNLPSolver.cpp:13-53): one process, one TNLP object, Ipopt
 * calling it from one thread — a second GPU is only reachable from inside the callback.  A group is one interval-sharded
 * engine per listed device (rank r computes a contiguous run of every phase's tiles, rank 0 also the endpoint rows); the
 * same device may be listed more than once.  Set-up calls (rpm_get_nlp_info, rpm_get_bounds_info, rpm_get_starting_point,
 * the structure passes, rpm_finalize_solution, the post-solve entry points) go to rpm_group_engine(g, 0): sizes and
 * layout are those of the unsharded problem.  Results are bit-identical to a single engine's (no reductions).  Calls on one
 * group are serial, as for an engine. */
#define RPM_GROUP_MAX 16
typedef struct rpm_group rpm_group;
int rpm_group_create(const rpm_problem_desc* desc, int n_devices, const int* device_ids, rpm_group** out);
void rpm_group_destroy(rpm_group* g);
const char* rpm_group_last_error(const rpm_group* g);      /* g == NULL: of the last failed rpm_group_create */
int rpm_group_size(const rpm_group* g);
rpm_engine* rpm_group_engine(rpm_group* g, int rank);
int rpm_group_device_init(rpm_group* g);                   /* optional: binds every engine to its device now */
int rpm_group_set_option(rpm_group* g, const char* key, int value);   /* rpm_set_option on every engine */
/* Host consumer — the TNLP callbacks (Core/LpopcIpopt.cpp:106-217) with every device on the data path: x, g, values are
 * the caller's arrays, page-locked once for all devices ("pin_host" is on in a group: the arrays must stay allocated until
 * rpm_group_destroy or rpm_group_set_option(g, "pin_host", 0)); every device reads x from them and stores ITS rows of g and,
 * by difference ("delta_values"), its runs of `values` straight into them over its own PCIe link; all devices are started
 * before any is waited for.  Objective, gradient and exact Hessian are evaluated by rank 0 (one small kernel each). */
int rpm_group_eval_f(rpm_group* g, int n, const double* x, int new_x, double* obj_value);
int rpm_group_eval_grad_f(rpm_group* g, int n, const double* x, int new_x, double* grad_f);
int rpm_group_eval_g(rpm_group* g, int n, const double* x, int new_x, int m, double* gvec);
int rpm_group_eval_jac_g(rpm_group* g, int n, const double* x, int new_x, int m, int nele_jac, int* iRow, int* jCol, double* values);
int rpm_group_eval_pair(rpm_group* g, int n, const double* x, int m, double* gvec, int nele_jac, double* values);
int rpm_group_eval_h(rpm_group* g, int n, const double* x, int new_x, double obj_factor, int m, const double* lambda, int new_lambda,
                     int nele_hess, int* iRow, int* jCol, double* values);
/* Device consumer on ONE device: d_x, d_g, d_values are arrays in the HBM of rank `home`'s device (instance strides as for
 * rpm_eval_pair_dev); the other ranks' tile kernels read x from and store their rows / runs into them directly over xGMI
 * (hipDeviceEnablePeerAccess) — no pack, no gather.  Blocking: returns when every rank is done. */
int rpm_group_eval_pair_dev(rpm_group* g, int home, const double* d_x, double* d_g, double* d_values);
/* All-gather: d_x[r], d_g[r], d_values[r] are full-size arrays in rank r's HBM; afterwards EVERY rank's g and values are
 * complete: each rank's tile kernel fills its rows / runs in its own arrays and one push kernel stores them into the same
 * places of every peer's arrays, one xGMI link per peer (a direct one-shot all-gather, not a ring).  Blocking. */
int rpm_group_allgather_pair_dev(rpm_group* g, const double* const* d_x, double* const* d_g, double* const* d_values);

/* ---- one process, several GPUs: the INSTANCES of a sweep dealt to the devices (the device solver of row f-2) ---------------
 * desc->n_instances independent NLPs of one transcription (the MPC sweep of BASELINE config 5), share r = instances
 * [B r / N, B (r + 1) / N) on device_ids[r] with an engine and a solver (rpm_ipm_*) of its own; the same device may be listed
 * more than once.  rpm_sweep_solve runs the shares side by side, a host thread each (the solver's loop blocks on its stream);
 * nothing crosses between devices, so every instance's result is what one engine holding all of them computes for it.  Options
 * are rpm_ipm_set_option's and go to every share; per-share objects (traces, kernel times, rpm_set_instance_constants with the
 * share's own instance numbers) through rpm_sweep_solver / rpm_sweep_engine.  Calls on one sweep are serial. */
typedef struct rpm_sweep rpm_sweep;
int rpm_sweep_create(const rpm_problem_desc* desc, int n_devices, const int* device_ids, rpm_sweep** out);
void rpm_sweep_destroy(rpm_sweep* s);
const char* rpm_sweep_last_error(const rpm_sweep* s);      /* s == NULL: of the last failed rpm_sweep_create */
int rpm_sweep_size(const rpm_sweep* s);
rpm_engine* rpm_sweep_engine(rpm_sweep* s, int share);
rpm_ipm* rpm_sweep_solver(rpm_sweep* s, int share);
int rpm_sweep_share(const rpm_sweep* s, int share, int* first_instance, int* n_instances);
int rpm_sweep_set_option(rpm_sweep* s, const char* key, double value);
int rpm_sweep_set_bounds(rpm_sweep* s, int instance, const double* x_l, const double* x_u);   /* instance: 0 .. B - 1 */
int rpm_sweep_solve(rpm_sweep* s, double* x, double* lambda, double* obj, int* status, int* iterations, double* kkt_error);
int rpm_sweep_get_stats(rpm_sweep* s, int* iterations, int* factorizations, int* trial_points);

#ifdef __cplusplus
}
#endif
#endif /* RPM_HIP_H_ */

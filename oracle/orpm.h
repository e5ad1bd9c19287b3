/*
 * orpm.h — CPU ORACLE for the lpopc hot path (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Plain-C restatement of the reference algorithm, each function citing the
 * reference file:line it follows (paths relative to /root/reference/Lpopc/src).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product (lpopc_amd/) never links, imports or calls it.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or known-answer
 * values and cannot be built in this image (it needs
 * Armadillo 5.300.4 and Ipopt 3.12.3 headers, neither vendored nor installed).
 * This oracle is therefore pinned only by (a) line-by-line restatement of the
 * reference sources and (b) reference-independent mathematical invariants
 * (tests/test_oracle_invariants.py).  Where the arithmetic order lives inside
 * Armadillo (accumulate / dot / prod), the published Armadillo 5.300.4 loop
 * order is restated and cited in the function's comment.
 *
 * It uses the product's problem *description* structs (include/rpm_hip.h) as its
 * input format, nothing else of the product.
 */
#ifndef ORPM_H_
#define ORPM_H_
#include "../include/rpm_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orpm orpm;

/* ---- vectorised user-callback ABI, mirrors Core/LpFunctionWrapper.h:12-69 ---- */
typedef struct {
  int phase_num;               /* 1-based */
  double initial_time;
  const double* initial_state; /* nx */
  double terminal_time;
  const double* terminal_state;
  int N, nx, nu, nq;
  const double* time;          /* N */
  const double* state;         /* N x nx, column-major */
  const double* control;       /* N x nu */
  const double* parameter;     /* nq */
} orpm_solcost;
typedef struct {
  int phase_num;
  int N, nx, nu, nq, nc;
  const double* time;
  const double* state;
  const double* control;
  const double* parameter;
} orpm_soldae;
typedef struct {
  int phase_num;
  double initial_time, terminal_time;
  int nx, nq, ne;
  const double* initial_state;
  const double* terminal_state;
  const double* parameter;
} orpm_solevent;
typedef struct {
  int left_phase_num, right_phase_num;
  int ipair;
  int nxl, nxr, nql, nqr, nlink;
  const double* left_state;
  const double* right_state;
  const double* left_parameter;
  const double* right_parameter;
} orpm_sollink;

typedef struct {
  void (*mayer)(const orpm_solcost*, const double* c, double* mayer);
  void (*lagrange)(const orpm_solcost*, const double* c, double* L);
  void (*dae)(const orpm_soldae*, const double* c, double* stateout, double* pathout);
  void (*event)(const orpm_solevent*, const double* c, double* eventout);
  void (*link)(const orpm_sollink*, const double* c, double* linkout);
  /* analytic derivatives (first-derive=analytic); NULL when the problem has none */
  void (*deriv_mayer)(const orpm_solcost*, const double* c, double* d /*1 x (2nx+2+nq)*/);
  void (*deriv_lagrange)(const orpm_solcost*, const double* c, double* d /*N x (nx+nu+1+nq)*/);
  void (*deriv_dae)(const orpm_soldae*, const double* c, double* dstate /*(N nx) x (nx+nu+1+nq)*/,
                    double* dpath /*(N nc) x (...)*/);
  void (*deriv_event)(const orpm_solevent*, const double* c, double* d /*ne x (2nx+2+nq)*/);
  void (*deriv_link)(const orpm_sollink*, const double* c, double* d /*nlink x (nxl+nql+nxr+nqr)*/);
} orpm_functions;

const orpm_functions* orpm_problem_functions(int problem_id);

/* ---- engine ------------------------------------------------------------------- */
orpm* orpm_create(const rpm_problem_desc* desc, char* err, int errlen);
void orpm_destroy(orpm* o);

/* CPU-baseline cost shapes (the numbers are bit-identical either way):
 *   0 "faithful-cost" (default): what lpopc itself pays — the COO sparse x dense loop with its column copies
 *     (SparseMatrix/LpSparseMatrix.cpp:127-155) and a Find(Doffdiag) scan per phase in GetWholeJacbi and again in
 *     GetPhaseJacbi on every eval_jac_g (Core/LpNLPWrapper.cpp:304, :686);
 *   1 "fair": the same algorithm and arithmetic order without those: D as dense rows per node (ascending columns, the
 *     order the COO loop produces per row), the Doffdiag value list found once per mesh. */
void orpm_set_cost_shape(orpm* o, int fair);

void orpm_get_nlp_info(const orpm* o, int* n, int* m, int* nnz_jac_g, int* nnz_h_lag);
void orpm_get_bounds_info(const orpm* o, double* x_l, double* x_u, double* g_l, double* g_u);
void orpm_get_starting_point(const orpm* o, double* x);
double orpm_eval_f(orpm* o, const double* x);
void orpm_eval_grad_f(orpm* o, const double* x, double* grad_f);
void orpm_eval_g(orpm* o, const double* x, double* g);
void orpm_jac_structure(orpm* o, int* iRow, int* jCol);
void orpm_eval_jac_g(orpm* o, const double* x, double* values);
void orpm_hess_structure(orpm* o, int* iRow, int* jCol);
void orpm_eval_h(orpm* o, const double* x, double obj_factor, const double* lambda, double* values);

void orpm_get_phase_sizes(const orpm* o, int phase, int* n_nodes, int* d_nnz, int* doff_nnz);
void orpm_get_phase_tables(const orpm* o, int phase, double* points, double* weights, int* d_rows,
                           int* d_cols, double* d_vals, double* diag_vals, int* doff_rows,
                           int* doff_cols, double* doff_vals);

/* Solution extraction, Nlp2OpConverter::Nlp2OpControl (Core/Nlp2OPConverter.cpp:13-196): per phase N+1 rows.
 * Any output may be NULL.  control/pathmult have the spline-extrapolated row at tau = +1 appended. */
void orpm_nlp2op(orpm* o, int phase, const double* x, const double* lambda, double* time, double* state,
                 double* control, double* costate, double* pathmult, double* hamiltonian, double* mayer_cost,
                 double* lagrange_cost);

/* Mesh-error estimate and ph refinement: SolutionErrorChecker::CheckSolutionDiffError
 * (Core/LpSolutionError.cpp:112-169) and PhMeshRefineAlg::RefineMesh (Core/LpPhMeshRefineAlg.cpp:12-100). */
int orpm_solution_error(orpm* o, int phase, const double* x, double* rel_err /* (N+K+1) x nx, column-major */);
int orpm_ph_refine(orpm* o, int phase, const double* x, double tol, int Nmin, int Nmax, double* new_mesh,
                   int* new_nodes, int* new_K, double* max_err_per_interval);
void orpm_bary_tables(int M, const double* data_x, int Nq, const double* xq, double* H, double* S, int* fix);
void orpm_inverse(int n, const double* A, double* inv);

/* hp-Liu mesh refinement, LiuHpMeshRefineAlg (Core/LpLiuHpMeshRefineAlg.cpp:12-709); stateful across meshes like the
 * reference object (mesh / state / mesh-point histories).  See oracle/orpm_hpliu.c for what is kept bug for bug. */
typedef struct orpm_hpliu orpm_hpliu;
orpm_hpliu* orpm_hpliu_create(int n_phases, double tol, int Nmax, double R);
void orpm_hpliu_destroy(orpm_hpliu* h);
int orpm_hpliu_refine(orpm_hpliu* h, orpm* o, const double* x, int cap, double* new_mesh, int* new_nodes, int* mesh_off,
                      int* nodes_off, int* new_K);
void orpm_hpliu_alj(int N, double* alj /* (N+1) x (N+1) */);

/* stand-alone table helpers (exposed for the invariant tests) */
void orpm_lgr_points(int n, double* x, double* w);                    /* RPMGenerator.cpp:253-291 */
void orpm_colloc_d(int M, const double* x, double* D /*(M-1) x M col-major*/); /* :107-130 */

#ifdef __cplusplus
}
#endif
#endif

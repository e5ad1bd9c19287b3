// rpm_ipm_device.hpp — what the two HIP translation units of row f-2 share: the parameter blocks of the interior-point
// kernels (rpm_ipm_kernels.hip) and their launchers, used by the solver loop and the C ABI (rpm_ipm_solver.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "rpm_ipm.hpp"

namespace rpm {

constexpr int IPM_W = 16;        // block width of the factorisation
constexpr int IPM_FMAX = 1024;   // filter entries kept per instance (the filter empties whenever mu changes; a long Delta-III phase at one mu adds hundreds)
constexpr int IPM_TRACE = 8;     // doubles per trace record: f, theta, mu, alpha, alpha_z, delta_w, E_0, backtracks
constexpr double IPM_INF = 1e19; // Ipopt's nlp_lower_bound_inf / nlp_upper_bound_inf
constexpr int IPM_LB_H = 6;      // limited-memory BFGS: Ipopt's limited_memory_max_history
constexpr int IPM_LB_PART = 80 * 16;   // doubles per instance for the partial sums of the limited-memory kernels: 80 sums x 16 waves
constexpr int IPM_LB_SMALL = 8 + 2 * (2 * IPM_LB_H) * (2 * IPM_LB_H) + 2 * (2 * IPM_LB_H);   // doubles of an instance's small record (rpm_ipm_lbfgs.hip)

struct IpmOpts {
  double tol = 1e-8, mu_init = 0.1, kappa_eps = 10.0, kappa_mu = 0.2, theta_mu = 1.5, tau_min = 0.99;
  double bound_push = 1e-2, bound_frac = 1e-2, kappa_sigma = 1e10, s_max = 100.0;
  double gamma_theta = 1e-5, gamma_phi = 1e-8, eta_phi = 1e-8, delta = 1.0, s_theta = 1.1, s_phi = 2.3, gamma_alpha = 0.05;
  double delta_c = 1e-9, delta_w_first = 1e-4, delta_w_min = 1e-20, delta_w_max = 1e40, kw_inc_first = 100.0, kw_inc = 8.0,
         kw_dec = 1.0 / 3.0;
  int max_iter = 3000, max_ls = 40;
  double acceptable_tol = 1e-6;      // Ipopt: "solved to acceptable level" after acceptable_iter consecutive such iterations
  int acceptable_iter = 15;
  int resto = 1, resto_max = 300;    // restoration phase (paper section 3.3) after a failed line search; its iteration limit
  double kappa_resto = 0.9, resto_rho = 1000.0, mult_reset = 1e3;
  double bound_relax = 1e-8;         // Ipopt's bound_relax_factor: finite bounds of free unknowns move out by this * max(1, |bound|)
  int max_soc = 4;                   // second-order correction steps per iteration (paper A-5.5 .. A-5.9)
  double kappa_soc = 0.99;
  int mu_adaptive = 1;               // 1: Ipopt's mu_strategy=adaptive with the LOQO oracle and the kkt-error globalisation (oracle/ipm_oracle.py)
  double mu_max_fact = 1e3, mu_red_fact = 0.9999, mu_init_factor = 0.8;
  double sigma_cap = 0.0;            // experiment: cap z/s in the KKT matrix (0 = off)
  int init_ls_mult = 0;              // 1: least-squares multipliers at the very first iterate too (Ipopt's default start)
  int nlp_scaling = 0;               // 1: Ipopt's gradient-based NLP scaling (its default; this solver's is off)
  double scal_gmax = 100.0, scal_min = 1e-8;   // nlp_scaling_max_gradient, nlp_scaling_min_value
  int ic_hot = 0;                    // 1: an iteration whose predecessor needed delta_w > 0 starts Algorithm IC at kw_dec * delta_w_last instead of 0
  double ic_hot_min = 1e-10;         //    (not Ipopt: saves the factorisation that fails at 0) as long as that value is at least this
  // Ipopt's unscaled termination thresholds, required beside the scaled E_0 <= tol (resp. acceptable_tol)
  double dual_inf_tol = 1.0, constr_viol_tol = 1e-4, compl_inf_tol = 1e-4;
  double acc_dual_inf_tol = 1e10, acc_constr_viol_tol = 1e-2, acc_compl_inf_tol = 1e-2;
};

struct IpmInst {
  double mu, tau, f, theta, lnsum, dinf, cinf, comp_max, comp_min, sum_lam, sum_z, err0;
  double delta_w, delta_w_last, alpha_max, alpha_z, alpha, alpha_min, dphi, phi, theta_max, theta_min;
  int status;   // 0 running, 1 converged, 6 converged to the acceptable level, 2 iteration limit, 3 line search failed (Ipopt would enter restoration), 4 inertia correction failed, 5 NaN/Inf
  int iter, nfilt, accepted, refactor, npos, nneg, nbad, ls, armijo, nzb, pad;
  int mode, resto_it, enter_resto, n_resto;   // mode 0 regular iteration, 2 restoration phase, 3 least-squares multipliers on leaving it
  int n_acc, skip_update;                     // consecutive iterations with E_0 <= acceptable_tol; this pass changed lambda only
  double th0, zeta;                           // restoration: infeasibility where it was entered, weight of the proximity term
  double mu_r, th_r, thr_max, thr_min, phi_r; // restoration: its own barrier parameter, infeasibility |c - p + n|_1, filter bounds, barrier objective
  int nrfilt, soc_on, soc_req, soc_p, use_soc, n_soc;   // restoration filter entries; second-order correction state
  double alpha_soc, az_soc, th_old_soc;
  double mu_max, refs[4];            // adaptive barrier update: upper bound of mu, KKT errors of the last accepted iterates
  int fixed_mode, nrefs;             // 0 = free mode (mu from the oracle every iteration), 1 = monotone rule until progress resumes
  int n_recalc, ic_hot;               // least-squares multipliers recomputed after a line search that failed at a feasible point (at most 3 times)
  long long dbg[8];   // phase clocks of the factorisation (builds with -DIPM_TIMING only)
};

// band + border storage of one instance (rpm_ipm.hpp): element (i, j), i >= j
struct KktGeom {
  int Nt, Nb, nb, b, CS;
  __device__ size_t at(int i, int j) const { return size_t(j) * CS + (i < Nb ? i - j : b + 1 + i - Nb); }
};
// one sub-problem of the factorisation (rpm_ipm.hpp KktSubHost): its geometry, where its block starts inside an instance's
// KKT storage, where its right-hand side starts inside an instance's vector
struct KktSub {
  KktGeom g;
  int roff;
  long long koff;
};
struct IpmDev {
  int B, n, m, ns, nv, Nt, Nb, nb, b, CS, nnz_jac, nnz_h;
  long long sg, sv, kstride;
  // plan tables
  const int *pos, *row_slack, *slack_row, *jac_dst, *hes_dst, *diag_dst, *slk_dst, *jt_ptr, *jt_ent, *jt_row;
  const int *hg_ptr, *hg_src, *hg_dst;   // Hessian entries grouped by storage slot
  int n_hg;
  // every structural slot of the KKT storage in ascending order (ipm_fill_kernel): as_ki = kind << 28 | index (kind 0 Hessian
  // slot hg i, 1 Jacobian entry k, 2 slack s, 3 diagonal of variable i — as_hg: the Hessian slot that shares it or -1 —,
  // 4 diagonal of constraint r); as_ptr[c] = first entry at or beyond chunk c (IPM_FILL_CHUNK doubles); as_nchunk 0 = not built
  const int *as_dst, *as_ki, *as_hg, *as_ptr;
  int as_nchunk;
  const int* long_cols;                  // Jacobian columns of more than 256 entries (a workgroup each in ipm_jt_lambda_kernel)
  int n_long;
  const double *gl, *gu;
  // per-instance state
  double *v, *vl, *vu, *zL, *zU, *lam, *dv, *dlam, *dzL, *dzU, *glag, *c, *rhs, *K, *filt;
  double *xe, *xt, *grad, *g, *jac, *hess, *obj, *gt, *objt;
  double *vR, *dr2;   // restoration: reference point and D_R^2 = 1 / max(1, |v_R|)^2
  double *vl0, *vu0;  // the caller's bounds; vl, vu are these moved out by bound_relax (ipm_init_kernel)
  double *pp, *nn, *zp, *zn, *dpp, *dnn, *dzp, *dzn;   // restoration: c(v) - p + n = 0, p, n >= 0, their multipliers and steps (m each)
  double* rfilt;      // restoration's own filter
  double* part;       // partial sums of the vector kernels that run several workgroups per instance (IPM_VEC_BLOCKS x IPM_VEC_PART per instance)
  int* tick;          // their arrival counters (one per instance, left at zero by the last workgroup)
  double *dv2, *dlam2, *dzL2, *dzU2, *csoc, *ct;       // second-order correction: candidate step, c_soc, c(trial point)
  double* trace;   // per instance trace_cap records of IPM_TRACE doubles (one per accepted step), or NULL
  int trace_cap;
  IpmInst* inst;
  int* cnt;     // [0] running, [1] to refactor, [2] line searches pending, [3] second-order corrections requested
  IpmOpts o;
  // factorisation sub-problems: one (the whole band + border matrix) or, with nested dissection, n_l1 interval blocks
  // followed by the separator system; pivot signs of every sub-problem land in piv[(instance * n_sub + sub) * 3 + {+,-,bad}]
  const KktSub* subs;
  int n_sub, n_l1, n_l2;    // n_sub = n_l1 + n_l2 + 1; n_l1 = 0: no dissection; n_l2 > 0: the separator system is cut into groups once more
  int* piv;
  const int *cg_ptr, *cg_src, *cg_dst;   // corner gather (level-1 Schur complements into level 2)
  const int *rg_ptr, *rg_src, *rg_dst;   // right-hand-side gather
  const int *rs_dst, *rs_src;            // solution scatter into the level-1 border work spaces
  const int* gap_pos;                    // positions of those work spaces (zeroed before the forward sweep)
  int n_cg, n_rg, n_rs, n_gap;
  const int *cg2_ptr, *cg2_src, *cg2_dst, *rg2_ptr, *rg2_src, *rg2_dst, *rs2_dst, *rs2_src;   // second stage (groups -> last level)
  int n_cg2, n_rg2, n_rs2;
  int n_cg_long, n_cg2_long;             // leading corner-gather destinations with >= 32 sources (a wave each)
  int max_sub_nt;                        // largest sub-problem order (right-hand side kept in LDS when it fits)
  int df_tiles;                         // tiles per level-1 sub-problem in df_map (>= IPM_DENSE_TILES)
  size_t l1_dense_lds;                   // > 0: level 1 runs kkt_factor_dense_kernel (every interval block fits its register tiles) with this much LDS
  int last_dense_corner;                 // 1: the last level's corner eliminated by kkt_factor_dense_kernel too (0, option "upper_dense" 2: by kkt_factor_kernel)
  size_t l2_dense_lds, last_dense_lds;   // the same for the groups of separators (partial, like level 1) and for the last level (that kernel then
                                         // eliminates the corner's block columns as well); option "upper_dense"
  // df_on: that kernel builds its interval block from the Jacobian / Hessian / diagonal terms itself instead of reading what
  // ipm_fill_kernel wrote (which then fills only the as_nlive chunks as_live[] of the storage that do not lie inside a level-1 block): the structural
  // slots of level-1 sub-problem s are df_ki / df_hg [df_ptr[3 s], df_ptr[3 s + 3]) (coded like as_ki / as_hg; Jacobian entries from
  // df_ptr[3 s], Hessian slots from df_ptr[3 s + 1], slack entries and diagonals from df_ptr[3 s + 2]), and lane l of register tile t
  // holds the entries numbered df_map[(s * IPM_DENSE_TILES + t) * 64 + l] (four 16-bit numbers, 1-based into that list, 0 = a
  // structural zero)
  int df_on;
  const int *df_ptr, *df_ki, *df_hg, *as_live;
  int as_nlive;
  const unsigned long long* df_map;
  // hessian-approximation = limited-memory (rpm_ipm_lbfgs.hip): no Hessian entries, sigma on the diagonal of x, low-rank part by Woodbury
  int rhs_mult;               // kkt_launch_solve: right-hand sides per instance in `rhs` (0 / 1: one; j-th of instance bi at row j * B + bi)
  int lb_on;
  double *lb_S, *lb_Y;        // B x IPM_LB_H x n pairs, oldest first
  double *lb_xprev;           // B x n: the iterate the stored gradient / Jacobian belong to
  double *lb_gold;            // B x nv: grad_x L(x_prev, lambda) with the CURRENT multipliers (first n of every row)
  // nlp_scaling: per instance the row factors sc (B x m), the objective factor sf (B), lambda o sc / sf for the Hessian call (B x m)
  int scal_on;
  double *sc, *sf, *lam_h;
  const int* jac_row;         // row of every Jacobian entry
  int nnz_var;                // entries of `jac` a later evaluation of the solve rewrites ([NL | LIN]; the constant block is written once)
  double *lb_small;           // B x IPM_LB_SMALL
  double *lb_part;            // B x IPM_LB_PART: the waves' shares of the sums of one phase (rpm_ipm_lbfgs.hip)
  double *lb_Z;               // 2 IPM_LB_H x B x Nt: K0^-1 E, column-major by column
};

constexpr int IPM_DENSE_SLOTS = 22, IPM_DENSE_TILE_WAVES = 7, IPM_DENSE_LDS_ROW = 18;   // kkt_factor_dense_kernel: tiles per wave, tile waves, doubles per LDS row
constexpr int IPM_DENSE_TILES = IPM_DENSE_SLOTS * IPM_DENSE_TILE_WAVES;
// kkt_factor_dense_kernel keeps the trailing IPM_DENSE_ROWS block rows of a block in registers; a block of up to IPM_DENSE_EARLY more
// eliminates its first ("early") block columns through the storage: their tiles are loaded, used and put back by the wave that owns them
// (4: three tiles per wave and early column wait in registers beside the resident ones; a fourth spills)
constexpr int IPM_DENSE_ROWS = 17, IPM_DENSE_EARLY = 4;
static_assert(IPM_DENSE_ROWS * (IPM_DENSE_ROWS + 1) / 2 <= IPM_DENSE_TILES, "resident tiles");
__host__ __device__ inline int ipm_dense_early(int block_rows) { return block_rows > IPM_DENSE_ROWS ? block_rows - IPM_DENSE_ROWS : 0; }
// number of tile (I, Kb), Kb <= I, of a block of `block_rows` block rows: the resident ones (Kb >= early columns) column by column
// from 0 — tile t sits in slot t / 7 of wave t % 7 —, the early ones after them
__host__ __device__ inline int ipm_dense_tile(int block_rows, int I, int Kb) {
  const int E = ipm_dense_early(block_rows), R = block_rows - E;
  if (Kb >= E) { const int kr = Kb - E; return kr * R - kr * (kr - 1) / 2 + I - Kb; }
  return R * (R + 1) / 2 + Kb * block_rows - Kb * (Kb - 1) / 2 + I - Kb;
}
__host__ __device__ inline int ipm_dense_tiles_of(int block_rows) { return block_rows * (block_rows + 1) / 2; }
constexpr int IPM_FILL_CHUNK = 4096;   // doubles of KKT storage one workgroup of ipm_fill_kernel zeroes and fills at a time (2048: 76 us, 4096: 70 us, 8192: 82 us on the metric problem)
constexpr int IPM_VEC_BLOCKS = 64;   // most workgroups per instance of a vector kernel
constexpr int IPM_VEC_PART = 24;     // doubles of partial results per workgroup
constexpr int IPM_MT = 8;   // most 16-row tiles per wave of the factorisation: block columns of up to 4 x 8 x 16 = 512 rows

// launchers (rpm_ipm_kernels.hip); all asynchronous on `st`
void ipm_launch_init(const IpmDev& D, const double* d_x0, hipStream_t st);
void ipm_launch_init_slack(const IpmDev& D, hipStream_t st);
void ipm_launch_pack_x(const IpmDev& D, hipStream_t st);
void ipm_launch_residual(const IpmDev& D, hipStream_t st);
void ipm_launch_assemble(const IpmDev& D, int nnz_max, hipStream_t st);          // zero + scatter + right-hand side
void ipm_launch_inertia(const IpmDev& D, hipStream_t st);
void ipm_launch_direction(const IpmDev& D, hipStream_t st);
void ipm_launch_trial(const IpmDev& D, hipStream_t st);
void ipm_launch_accept(const IpmDev& D, hipStream_t st);
void ipm_launch_update(const IpmDev& D, hipStream_t st);
void ipm_launch_soc_rhs(const IpmDev& D, hipStream_t st);          // right-hand side of the second-order correction
void ipm_launch_soc_direction(const IpmDev& D, hipStream_t st);    // its step and step lengths
// Ipopt's gradient-based NLP scaling (option nlp_scaling)
void ipm_launch_scaling_factors(const IpmDev& D, hipStream_t st);   // sc, sf from D.grad / D.jac at the starting point (unscaled)
void ipm_launch_scale(const IpmDev& D, double* g, double* jac, int jac0, int jac1, double* obj, double* grad, hipStream_t st);   // in place; null = leave
void ipm_launch_scale_lambda(const IpmDev& D, hipStream_t st);       // lam_h = lam o sc / sf
void ipm_launch_scale_hessian(const IpmDev& D, hipStream_t st);      // hess *= sf
void ipm_launch_unscale_lambda(const IpmDev& D, double* out, hipStream_t st);   // out = lam o sc / sf
// limited-memory BFGS (rpm_ipm_lbfgs.hip)
void lb_launch_reset(const IpmDev& D, hipStream_t st);
void lb_launch_update(const IpmDev& D, hipStream_t st);               // after the residual kernel of an iteration
void lb_launch_columns_and_solve(const IpmDev& D, hipStream_t st);   // Z <- K0^-1 E, every column in one pass
void lb_launch_small(const IpmDev& D, hipStream_t st);                // C = M - E'Z, LU
void lb_launch_correct(const IpmDev& D, int check_status, hipStream_t st);   // Woodbury correction of the solution in D.rhs
void ipm_launch_jt_lambda_into(const IpmDev& D, double* out, hipStream_t st);   // grad f + A'lambda of the running instances into out (B x nv)
// factorisation / substitution of every running instance; tiles_per_wave 4 or IPM_MT
size_t kkt_factor_lds_bytes(const IpmPlan& p);
hipError_t kkt_factor_prepare(int tiles_per_wave, size_t lds_bytes);
size_t kkt_factor_dense_lds_bytes(int block_rows);
int kkt_factor_dense_max_block_rows();
hipError_t kkt_factor_dense_prepare(size_t lds_bytes);
// factor every running instance that asks for it / solve in place in D.rhs: the band + border matrix, or level 1 ->
// corner gather -> level 2 and forward -> gather -> level 2 -> scatter -> backward with nested dissection
void kkt_launch_factor(const IpmDev& D, int tiles_per_wave, size_t lds_bytes, hipStream_t st);
// check_status: 0 every instance, 1 the running ones, 2 the running ones that asked for a second-order correction
// forward_done: D.rhs is the right-hand side the factorisation just ran over (kkt_level1_fused: its level-1 forward sweep is done)
void kkt_launch_solve(const IpmDev& D, int check_status, hipStream_t st, int forward_done = 0);
int kkt_level1_fused(const IpmDev& D);

}  // namespace rpm

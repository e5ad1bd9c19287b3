// rpm_ipm_kernels.hip — row f-2: batched primal-dual interior-point iterations with every iterate, multiplier, KKT
// matrix and factor resident in HBM (see rpm_ipm.hpp for what is restated and what is not).  One workgroup per
// instance for the vector work and for the band + border LDL^T; the NLP callbacks are the engine's own batched launches.
//
// Per iteration (Waechter & Biegler 2006, the equation numbers below are that paper's):
//   residuals, optimality error E_0 / E_mu (5), barrier update (7), tau (8)      ipm_residual_kernel
//   W = eval_h(x, 1, lambda); K = [[W + Sigma + dw I, A^T], [A, -dc I]] (13)     ipm_assemble_kernel
//   LDL^T without pivoting + inertia check / correction (Algorithm IC)           kkt_factor_kernel, ipm_inertia_kernel
//   direction, dz (12), fraction to the boundary (15), alpha_min (23)            kkt_solve_kernel, ipm_direction_kernel
//   filter line search (18)-(20), (22)                                           ipm_trial_kernel, ipm_accept_kernel
//   step, multiplier reset (16), filter update                                   ipm_update_kernel
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>

#include "rpm_device_internal.hpp"
#include "rpm_ipm.hpp"

namespace rpm {

constexpr int IPM_W = 16;        // block width of the factorisation
constexpr int IPM_FMAX = 256;    // filter entries kept per instance
constexpr int IPM_TRACE = 8;     // doubles per trace record: f, theta, mu, alpha, alpha_z, delta_w, E_0, backtracks
constexpr double IPM_INF = 1e19; // Ipopt's nlp_lower_bound_inf / nlp_upper_bound_inf

struct IpmOpts {
  double tol = 1e-8, mu_init = 0.1, kappa_eps = 10.0, kappa_mu = 0.2, theta_mu = 1.5, tau_min = 0.99;
  double bound_push = 1e-2, bound_frac = 1e-2, kappa_sigma = 1e10, s_max = 100.0;
  double gamma_theta = 1e-5, gamma_phi = 1e-8, eta_phi = 1e-8, delta = 1.0, s_theta = 1.1, s_phi = 2.3, gamma_alpha = 0.05;
  double delta_c = 1e-8, delta_w_first = 1e-4, delta_w_min = 1e-20, delta_w_max = 1e40, kw_inc_first = 100.0, kw_inc = 8.0,
         kw_dec = 1.0 / 3.0;
  int max_iter = 3000, max_ls = 40;
  double acceptable_tol = 1e-6;      // Ipopt: "solved to acceptable level" after acceptable_iter consecutive such iterations
  int acceptable_iter = 15;
  int resto = 1, resto_max = 60;     // Gauss-Newton feasibility restoration after a failed line search
  double kappa_resto = 0.9;
};

struct IpmInst {
  double mu, tau, f, theta, lnsum, dinf, cinf, comp_max, comp_min, sum_lam, sum_z, err0;
  double delta_w, delta_w_last, alpha_max, alpha_z, alpha, alpha_min, dphi, phi, theta_max, theta_min;
  int status;   // 0 running, 1 converged, 6 converged to the acceptable level, 2 iteration limit, 3 line search failed (Ipopt would enter restoration), 4 inertia correction failed, 5 NaN/Inf
  int iter, nfilt, accepted, refactor, npos, nneg, nbad, ls, armijo, nzb, pad;
  int mode, resto_it, enter_resto, n_resto;   // mode 1: feasibility restoration
  int n_acc, pad2;                            // consecutive iterations with E_0 <= acceptable_tol
  double th0, zeta, psi, slope;
  long long dbg[8];   // phase clocks of the factorisation (builds with -DIPM_TIMING only)
};

struct IpmDev {
  int B, n, m, ns, nv, Nt, Nb, nb, b, CS, nnz_jac, nnz_h;
  long long sg, sv, kstride;
  // plan tables
  const int *pos, *row_slack, *slack_row, *jac_dst, *hes_dst, *diag_dst, *slk_dst, *jt_ptr, *jt_ent, *jt_row;
  const double *gl, *gu;
  // per-instance state
  double *v, *vl, *vu, *zL, *zU, *lam, *dv, *dlam, *dzL, *dzU, *glag, *c, *rhs, *K, *filt;
  double *xe, *xt, *grad, *g, *jac, *hess, *obj, *gt, *objt;
  double *vR, *dr2;   // restoration: reference point and D_R^2 = 1 / max(1, |v_R|)^2
  double* trace;   // per instance trace_cap records of IPM_TRACE doubles (one per accepted step), or NULL
  int trace_cap;
  IpmInst* inst;
  int* cnt;     // [0] running, [1] to refactor, [2] line searches pending
  IpmOpts o;
};

// ------------------------------------------------------------------------------------------------ helpers
__device__ inline double block_red(double v, int kind, double* sh) {   // 0 sum, 1 max, 2 min; result on every thread
  for (int o = 32; o; o >>= 1) {
    const double w = __shfl_down(v, o);
    v = kind == 0 ? v + w : (kind == 1 ? fmax(v, w) : fmin(v, w));
  }
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  double r = sh[0];
  for (int i = 1; i < int(blockDim.x >> 6); ++i) r = kind == 0 ? r + sh[i] : (kind == 1 ? fmax(r, sh[i]) : fmin(r, sh[i]));
  return r;
}
__device__ inline bool has_lo(double l, double u) { return l > -IPM_INF && l != u; }
__device__ inline bool has_up(double l, double u) { return u < IPM_INF && l != u; }

// ------------------------------------------------------------------------------------------------ start
// x pushed into the interior of its bounds (Ipopt 3.12 bound_push / bound_frac, paper section 3.6), z = 1, lambda = 0
__global__ __launch_bounds__(256) void ipm_init_kernel(IpmDev D, const double* x0) {
  const int bi = blockIdx.x;
  double* v = D.v + size_t(bi) * D.nv;
  const double *vl = D.vl + size_t(bi) * D.nv, *vu = D.vu + size_t(bi) * D.nv;
  for (int i = threadIdx.x; i < D.n; i += blockDim.x) {
    double x = x0[size_t(bi) * D.n + i];
    const double l = vl[i], u = vu[i];
    if (l == u) x = l;
    else {
      const bool lo = l > -IPM_INF, up = u < IPM_INF;
      if (lo) {
        const double p = up ? fmin(D.o.bound_push * fmax(1.0, fabs(l)), D.o.bound_frac * (u - l)) : D.o.bound_push * fmax(1.0, fabs(l));
        x = fmax(x, l + p);
      }
      if (up) {
        const double p = lo ? fmin(D.o.bound_push * fmax(1.0, fabs(u)), D.o.bound_frac * (u - l)) : D.o.bound_push * fmax(1.0, fabs(u));
        x = fmin(x, u - p);
      }
    }
    v[i] = x;
    D.zL[size_t(bi) * D.nv + i] = has_lo(l, u) ? 1.0 : 0.0;
    D.zU[size_t(bi) * D.nv + i] = has_up(l, u) ? 1.0 : 0.0;
  }
  for (int r = threadIdx.x; r < D.m; r += blockDim.x) D.lam[size_t(bi) * D.m + r] = 0.0;
  if (threadIdx.x == 0) {
    IpmInst& S = D.inst[bi];
    S = IpmInst{};
    S.mu = D.o.mu_init;
  }
}
// slacks start at g(x0), pushed inside [g_l, g_u] the same way
__global__ __launch_bounds__(256) void ipm_init_slack_kernel(IpmDev D) {
  const int bi = blockIdx.x;
  for (int s = threadIdx.x; s < D.ns; s += blockDim.x) {
    const int r = D.slack_row[s];
    const double l = D.gl[r], u = D.gu[r];
    double x = D.g[size_t(bi) * D.sg + r];
    const bool lo = l > -IPM_INF, up = u < IPM_INF;
    if (lo) {
      const double p = up ? fmin(D.o.bound_push * fmax(1.0, fabs(l)), D.o.bound_frac * (u - l)) : D.o.bound_push * fmax(1.0, fabs(l));
      x = fmax(x, l + p);
    }
    if (up) {
      const double p = lo ? fmin(D.o.bound_push * fmax(1.0, fabs(u)), D.o.bound_frac * (u - l)) : D.o.bound_push * fmax(1.0, fabs(u));
      x = fmin(x, u - p);
    }
    const size_t o = size_t(bi) * D.nv + D.n + s;
    D.v[o] = x;
    D.vl[o] = l;
    D.vu[o] = u;
    D.zL[o] = lo ? 1.0 : 0.0;
    D.zU[o] = up ? 1.0 : 0.0;
  }
}
__global__ void ipm_pack_x_kernel(IpmDev D) {
  const int bi = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < D.n) D.xe[size_t(bi) * D.n + i] = D.v[size_t(bi) * D.nv + i];
}

// ------------------------------------------------------------------------------------------------ residuals, E_mu, mu
__global__ __launch_bounds__(256) void ipm_residual_kernel(IpmDev D) {
  __shared__ double sh[4];
  const int bi = blockIdx.x, t = threadIdx.x;
  IpmInst& S = D.inst[bi];
  if (S.status != 0) return;
  const double *v = D.v + size_t(bi) * D.nv, *vl = D.vl + size_t(bi) * D.nv, *vu = D.vu + size_t(bi) * D.nv;
  const double *zL = D.zL + size_t(bi) * D.nv, *zU = D.zU + size_t(bi) * D.nv, *lam = D.lam + size_t(bi) * D.m;
  const double *jac = D.jac + size_t(bi) * D.sv, *g = D.g + size_t(bi) * D.sg, *grad = D.grad + size_t(bi) * D.n;
  double* glag = D.glag + size_t(bi) * D.nv;
  double dinf = 0, cinf = 0, th1 = 0, cmax = 0, cmin = 1e300, sl = 0, sz = 0, ln = 0, bad = 0, nzb = 0;
  // pass 1: constraint values and what does not depend on the multipliers
  double csq = 0, qd = 0;
  for (int r = t; r < D.m; r += blockDim.x) {
    const int s = D.row_slack[r];
    const double cr = s < 0 ? g[r] - D.gl[r] : g[r] - v[D.n + s];
    D.c[size_t(bi) * D.m + r] = cr;
    if (!(fabs(cr) < 1e300)) bad = 1;
    cinf = fmax(cinf, fabs(cr));
    th1 += fabs(cr);
    csq += cr * cr;
  }
  cinf = block_red(cinf, 1, sh); th1 = block_red(th1, 0, sh);
  __shared__ int verdict;       // restoration: 0 stay, 1 back to the regular iteration, 2 stop
  const int mode_in = S.mode;
  if (mode_in == 1) {
    const double *vR = D.vR + size_t(bi) * D.nv, *dr2 = D.dr2 + size_t(bi) * D.nv;
    for (int i = t; i < D.nv; i += blockDim.x) {
      const double l = vl[i], u = vu[i];
      if (l == u) continue;
      if (l > -IPM_INF) ln += log(v[i] - l);
      if (u < IPM_INF) ln += log(u - v[i]);
      const double dd = v[i] - vR[i];
      qd += dr2[i] * dd * dd;
    }
    csq = block_red(csq, 0, sh); ln = block_red(ln, 0, sh); qd = block_red(qd, 0, sh); bad = block_red(bad, 1, sh);
    if (t == 0) {
      const IpmOpts& o = D.o;
      const double f = D.obj[bi], phi = f - S.mu * ln;
      verdict = 0;
      if (bad != 0 || !(fabs(f) < 1e300) || !(fabs(ln) < 1e300)) { S.status = 5; verdict = 2; }
      else {
        bool back = S.resto_it > 0 && th1 <= o.kappa_resto * S.th0 && th1 <= S.theta_max;
        const double* F = D.filt + size_t(bi) * 2 * IPM_FMAX;
        for (int k = 0; back && k < S.nfilt; ++k)
          if (th1 >= F[2 * k] && phi >= F[2 * k + 1]) back = false;
        if (back) verdict = 1;
        else if (S.resto_it >= o.resto_max || S.iter >= o.max_iter) { S.status = S.iter >= o.max_iter ? 2 : 3; verdict = 2; }
        else {
          S.f = f; S.theta = th1; S.lnsum = ln; S.cinf = cinf;
          S.psi = 0.5 * csq + 0.5 * S.zeta * qd - S.mu * ln;
          S.refactor = 1;
          S.delta_w = 0.0;
          atomicAdd(&D.cnt[0], 1);
        }
      }
    }
    __syncthreads();
    if (verdict != 1) return;
    // back to the regular iteration: lambda = 0, bound multipliers clipped (as after Ipopt's restoration)
    for (int r = t; r < D.m; r += blockDim.x) D.lam[size_t(bi) * D.m + r] = 0.0;
    for (int i = t; i < D.nv; i += blockDim.x) {
      const double l = vl[i], u = vu[i];
      if (l == u) continue;
      const size_t o2 = size_t(bi) * D.nv + i;
      if (l > -IPM_INF) { const double sl2 = v[i] - l; D.zL[o2] = fmax(fmin(fmin(D.zL[o2], 1e3), D.o.kappa_sigma * S.mu / sl2), S.mu / (D.o.kappa_sigma * sl2)); }
      if (u < IPM_INF) { const double su2 = u - v[i]; D.zU[o2] = fmax(fmin(fmin(D.zU[o2], 1e3), D.o.kappa_sigma * S.mu / su2), S.mu / (D.o.kappa_sigma * su2)); }
    }
    if (t == 0) { S.mode = 0; S.n_resto += 1; }
    __syncthreads();
    ln = 0; bad = 0;
  }
  // pass 2: gradient of the Lagrangian, complementarity products
  for (int i = t; i < D.nv; i += blockDim.x) {
    double acc;
    if (i < D.n) {
      acc = grad[i];
      for (int q = D.jt_ptr[i]; q < D.jt_ptr[i + 1]; ++q) acc += jac[D.jt_ent[q]] * lam[D.jt_row[q]];
    } else {
      acc = -lam[D.slack_row[i - D.n]];
    }
    glag[i] = acc;
    const double l = vl[i], u = vu[i];
    if (l != u) {
      dinf = fmax(dinf, fabs(acc - zL[i] + zU[i]));
      if (!(fabs(acc) < 1e300)) bad = 1;
      if (l > -IPM_INF) {
        const double d = v[i] - l, pr = zL[i] * d;
        cmax = fmax(cmax, pr); cmin = fmin(cmin, pr); sz += zL[i]; ln += log(d); nzb += 1;
      }
      if (u < IPM_INF) {
        const double d = u - v[i], pr = zU[i] * d;
        cmax = fmax(cmax, pr); cmin = fmin(cmin, pr); sz += zU[i]; ln += log(d); nzb += 1;
      }
    }
  }
  for (int r = t; r < D.m; r += blockDim.x) sl += fabs(lam[r]);
  dinf = block_red(dinf, 1, sh);
  cmax = block_red(cmax, 1, sh); cmin = block_red(cmin, 2, sh); sl = block_red(sl, 0, sh); sz = block_red(sz, 0, sh);
  ln = block_red(ln, 0, sh); bad = block_red(bad, 1, sh); nzb = block_red(nzb, 0, sh);
  if (t != 0) return;
  const IpmOpts& o = D.o;
  S.f = D.obj[bi]; S.theta = th1; S.lnsum = ln; S.dinf = dinf; S.cinf = cinf; S.comp_max = cmax; S.comp_min = cmin;
  S.sum_lam = sl; S.sum_z = sz; S.nzb = int(nzb);
  if (!(fabs(S.f) < 1e300) || !(fabs(ln) < 1e300)) bad = 1;
  const double sd = fmax(o.s_max, (sl + sz) / fmax(1.0, double(D.m) + nzb)) / o.s_max;   // (6)
  const double sc = nzb > 0 ? fmax(o.s_max, sz / nzb) / o.s_max : 1.0;
  S.err0 = fmax(fmax(dinf / sd, cinf), nzb > 0 ? cmax / sc : 0.0);
  if (bad != 0) { S.status = 5; return; }
  if (S.err0 <= o.tol) { S.status = 1; return; }
  S.n_acc = S.err0 <= o.acceptable_tol ? S.n_acc + 1 : 0;
  if (o.acceptable_iter > 0 && S.n_acc >= o.acceptable_iter) { S.status = 6; return; }
  if (S.iter >= o.max_iter) { S.status = 2; return; }
  if (S.iter == 0) {
    S.theta_max = 1e4 * fmax(1.0, th1);
    S.theta_min = 1e-4 * fmax(1.0, th1);
  }
  const double mu_min = o.tol / 10.0;
  double mu = S.mu;
  for (int guard = 0; guard < 64; ++guard) {
    const double comp = nzb > 0 ? fmax(fabs(cmax - mu), fabs(cmin - mu)) : 0.0;
    const double emu = fmax(fmax(dinf / sd, cinf), comp / sc);
    if (!(emu <= o.kappa_eps * mu) || mu <= mu_min) break;
    mu = fmax(mu_min, fmin(o.kappa_mu * mu, pow(mu, o.theta_mu)));   // (7)
    S.nfilt = 0;
  }
  S.mu = mu;
  S.tau = fmax(o.tau_min, 1.0 - mu);   // (8)
  S.phi = S.f - mu * ln;
  S.refactor = 1;
  S.delta_w = 0.0;
  atomicAdd(&D.cnt[0], 1);
}

// ------------------------------------------------------------------------------------------------ KKT matrix + rhs
__global__ void ipm_zero_kernel(IpmDev D) {
  const int bi = blockIdx.y;
  const IpmInst& S = D.inst[bi];
  if (S.status != 0 || !S.refactor) return;
  double2* K = reinterpret_cast<double2*>(D.K + size_t(bi) * D.kstride);
  const long long n2 = (long long)D.Nt * D.CS / 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long long)gridDim.x * blockDim.x)
    K[i] = make_double2(0.0, 0.0);
}
__global__ void ipm_assemble_kernel(IpmDev D) {
  const int bi = blockIdx.y;
  const IpmInst& S = D.inst[bi];
  if (S.status != 0 || !S.refactor) return;
  double* K = D.K + size_t(bi) * D.kstride;
  const double *v = D.v + size_t(bi) * D.nv, *vl = D.vl + size_t(bi) * D.nv, *vu = D.vu + size_t(bi) * D.nv;
  const double *zL = D.zL + size_t(bi) * D.nv, *zU = D.zU + size_t(bi) * D.nv;
  const int stride = gridDim.x * blockDim.x, t0 = blockIdx.x * blockDim.x + threadIdx.x;
  const bool resto = S.mode == 1;      // restoration: W = zeta D_R^2 + mu / s^2, -I in the constraint block, no Hessian
  if (!resto)
    for (int k = t0; k < D.nnz_h; k += stride)
      if (D.hes_dst[k] >= 0) unsafeAtomicAdd(&K[D.hes_dst[k]], D.hess[size_t(bi) * D.nnz_h + k]);
  for (int k = t0; k < D.nnz_jac; k += stride)
    if (D.jac_dst[k] >= 0) K[D.jac_dst[k]] = D.jac[size_t(bi) * D.sv + k];
  for (int s = t0; s < D.ns; s += stride) K[D.slk_dst[s]] = -1.0;
  double* rhs = D.rhs + size_t(bi) * D.Nt;
  for (int i = t0; i < D.nv; i += stride) {
    const double l = vl[i], u = vu[i];
    double diag = 1.0, r = 0.0;
    if (l != u && !resto) {
      diag = S.delta_w;
      r = D.glag[size_t(bi) * D.nv + i];
      if (l > -IPM_INF) { const double d = v[i] - l; diag += zL[i] / d; r -= S.mu / d; }
      if (u < IPM_INF) { const double d = u - v[i]; diag += zU[i] / d; r += S.mu / d; }
    } else if (l != u) {
      const double w2 = S.zeta * D.dr2[size_t(bi) * D.nv + i];
      diag = S.delta_w + w2;
      r = w2 * (v[i] - D.vR[size_t(bi) * D.nv + i]);
      if (l > -IPM_INF) { const double d = v[i] - l; diag += S.mu / (d * d); r -= S.mu / d; }
      if (u < IPM_INF) { const double d = u - v[i]; diag += S.mu / (d * d); r += S.mu / d; }
    }
    unsafeAtomicAdd(&K[D.diag_dst[i]], diag);
    rhs[D.pos[i]] = -r;
  }
  for (int r = t0; r < D.m; r += stride) {
    K[D.diag_dst[D.nv + r]] = resto ? -1.0 : -D.o.delta_c;
    rhs[D.pos[D.nv + r]] = -D.c[size_t(bi) * D.m + r];
  }
}

// ------------------------------------------------------------------------------------------------ band + border LDL^T
// Storage of one instance: column j holds rows j .. j+b of the band (slot i-j) and the nb border rows (slot b+1+i-Nb).
// IPM_W columns at a time: the diagonal block is factored in LDS, each panel row is solved by the thread that owns it.
// No pivoting: with dw large enough and dc > 0 the matrix is symmetric quasi-definite, whose LDL^T exists for every
// ordering (Vanderbei 1995); the signs of D give the inertia Algorithm IC asks for.
struct KktGeom {
  int Nt, Nb, nb, b, CS;
  __device__ size_t at(int i, int j) const { return size_t(j) * CS + (i < Nb ? i - j : b + 1 + i - Nb); }
};
__device__ inline void block_range(const KktGeom& G, int J0, int* J1, int* nrb, int* nr) {
  if (J0 < G.Nb) {
    *J1 = min(J0 + IPM_W, G.Nb);
    const int last = min(*J1 - 1 + G.b, G.Nb - 1);
    *nrb = max(last - *J1 + 1, 0);
    *nr = *nrb + G.nb;
  } else {
    *J1 = min(J0 + IPM_W, G.Nt);
    *nrb = 0;
    *nr = G.Nt - *J1;
  }
}
__device__ inline int panel_row(const KktGeom& G, int J0, int J1, int nrb, int q) {
  return J0 >= G.Nb ? J1 + q : (q < nrb ? J1 + q : G.Nb + (q - nrb));
}

// Left-looking over the band: block column J (IPM_W = 16 columns) gathers the contributions of the b columns before it,
//     A(rows, J..J+15)^T  -=  T^T (16 x k) . L(rows, k)^T (k x 16 rows),      T[k][c] = d_k L(J+c, k),
// as v_mfma_f64_16x16x4_f64 products: one 16-row tile of the matrix per accumulator, the tiles of a block column dealt to
// the 4 waves, T staged in LDS (one conflict-free 8-byte read per lane and 4 columns), L streamed from HBM exactly where
// it is non-zero (a tile starts at its first in-band column).  Every factor entry is read ~b/16 times and written once
// (the right-looking form re-writes the whole (b + nb)^2 window per block column).  The panel solve is a second
// matrix product, Y^T = L11^-1 A^T, with the accumulators fed back as the B operand.  The border x border corner lives
// in LDS, is updated right-looking and factored there.  Operand maps (cdna_hip_programming.md §3): A: lane l holds
// A[l&15][l>>4], B: B[l>>4][l&15], C/D: row (l>>4) + 4 reg, column l&15.
#ifndef IPM_LB
#define IPM_LB 3   // waves per SIMD the 4-tile factorisation is compiled for
#endif
constexpr int IPM_MT = 8;   // most 16-row tiles per wave: block columns of up to 4 x 8 x 16 = 512 rows
template <int MT>           // 16-row tiles per wave: 4 (block columns of up to 256 rows, 4 workgroups per CU) or 8
__global__ __launch_bounds__(256, MT == 4 ? IPM_LB : 1) void kkt_factor_kernel(double* Kall, long long kstride, KktGeom G, IpmInst* inst) {
  typedef double d4 __attribute__((ext_vector_type(4)));
  constexpr int W = IPM_W;
  const int bi = blockIdx.x, t = threadIdx.x, nt = blockDim.x;
  IpmInst& S = inst[bi];
  if (S.status != 0 || !S.refactor) return;
  double* K = Kall + size_t(bi) * kstride;
  extern __shared__ double lds[];
  double* T = lds;                              // (b + 8) x W
  double* Dg = T + size_t(G.b + 8) * W;         // W x (W + 1)
  double* Mi = Dg + W * (W + 1);                // W x W: L11^-1, row-major
  double* invd = Mi + W * W;                    // W
  double* BL = invd + W;                        // nb x W: L of the border rows in the current block column
  double* BY = BL + size_t(G.nb) * W;           // nb x W: L D
  double* C = BY + size_t(G.nb) * W;            // nb x nb: the corner (lower triangle used)
  __shared__ int cnt[3];
  if (t < 3) cnt[t] = 0;
  int npos = 0, nneg = 0, nbad = 0;
  const int nb = G.nb;
  const int wv = t >> 6, lr = t & 15, lq = (t & 63) >> 4;
#ifdef IPM_TIMING
  long long tc[6] = {0, 0, 0, 0, 0, 0}, t_prev = wall_clock64();
#define IPM_TICK(i) do { __syncthreads(); const long long _n = wall_clock64(); tc[i] += _n - t_prev; t_prev = _n; } while (0)
#else
#define IPM_TICK(i)
#endif
  for (int idx = t; idx < nb * nb; idx += nt) {
    const int r = idx / nb, c = idx % nb;
    C[idx] = r >= c ? K[G.at(G.Nb + r, G.Nb + c)] : 0.0;
  }
  for (int J0 = 0; J0 < G.Nb;) {
    const int J1 = min(J0 + W, G.Nb), w = J1 - J0;
    const int kbase = max(J0 - G.b, 0), nk = J0 - kbase, ngrp = (nk + 3) >> 2;
    const int nrb = max(min(J1 - 1 + G.b, G.Nb - 1) - J1 + 1, 0), rows = w + nrb + nb, ntile = (rows + 15) >> 4;
#pragma unroll 4
    for (int idx = t; idx < (ngrp + 1) * 4 * W; idx += nt) {
      const int kk = idx / W, c = idx % W, k = kbase + kk, j = J0 + c;
      T[idx] = (kk < nk && c < w && j - k <= G.b) ? K[G.at(j, k)] * K[G.at(k, k)] : 0.0;
    }
    __syncthreads();
    IPM_TICK(0);
    d4 acc[MT];
    const double* lp[MT];       // &L(r, kbase) of this lane's row in tile i: in-band rows walk columns with stride CS - 1, border rows CS
    int lstep[MT], kfirst[MT], gfirst[MT], rrow[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int q0 = (wv + 4 * i) * 16, q = q0 + lr;
      const bool valid = q < rows;
      const int r = q < w ? J0 + q : (q - w < nrb ? J1 + (q - w) : G.Nb + (q - w - nrb));
      const bool border = r >= G.Nb;
      rrow[i] = valid ? r : -1;
      kfirst[i] = valid ? (border ? 0 : max(r - G.b - kbase, 0)) : (1 << 30);
      lp[i] = K + (valid ? G.at(r, kbase) : 0);
      lstep[i] = border ? G.CS : G.CS - 1;
      // the tile's first group of 4 columns with anything stored: from its first row when the whole tile is inside the band
      const int r_first = q0 < w ? J0 + q0 : J1 + (q0 - w);
      gfirst[i] = q0 >= rows ? (1 << 30) : ((q0 + 15 >= w + nrb) ? 0 : (max(r_first - G.b - kbase, 0) >> 2));
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = lq + 4 * g;
        acc[i][g] = (valid && c < w && J0 + c <= r && (border || r - (J0 + c) <= G.b)) ? K[G.at(r, J0 + c)] : 0.0;
      }
    }
    {
      // two column groups of L in flight per tile while the previous two are multiplied
      auto ld = [&](int i, int kg) -> double {
        const int kk = 4 * kg + lq;
        return (kg >= gfirst[i] && kk >= kfirst[i] && kk < nk) ? lp[i][size_t(kk) * lstep[i]] : 0.0;
      };
      double b0[MT], b1[MT];
#pragma unroll
      for (int i = 0; i < MT; ++i) { b0[i] = ld(i, 0); b1[i] = ld(i, 1); }
      for (int kg = 0; kg < ngrp; kg += 2) {
        const double a0 = -T[(4 * kg + lq) * W + lr], a1 = -T[(4 * kg + 4 + lq) * W + lr];
        double n0[MT], n1[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) { n0[i] = ld(i, kg + 2); n1[i] = ld(i, kg + 3); }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          if (kg + 1 < gfirst[i]) continue;
          acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0[i], acc[i], 0, 0, 0);
          acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1[i], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) { b0[i] = n0[i]; b1[i] = n1[i]; }
      }
    }
    IPM_TICK(1);
    if (wv == 0) {              // tile 0 holds the diagonal block (rows q < w): its LDL^T in the registers of 16 lanes
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = lq + 4 * g;
        if (lr < w && c <= lr) Dg[lr * (W + 1) + c] = acc[0][g];
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      double row[W], inv[W];    // lane (l & 15) = row of the block and of L11^-1 (the same eliminations applied to I)
#pragma unroll
      for (int c = 0; c < W; ++c) {
        row[c] = (lr < w && c <= lr) ? Dg[lr * (W + 1) + c] : (c == lr ? 1.0 : 0.0);
        inv[c] = c == lr ? 1.0 : 0.0;
      }
#pragma unroll
      for (int k = 0; k < W; ++k) {
        // all cross-lane reads of the step first (they are independent), then the arithmetic
        const double dk = __shfl(row[k], k, 16);                   // pivot: row k's own diagonal
        double ajk[W], mkj[W];
#pragma unroll
        for (int j = k + 1; j < W; ++j) ajk[j] = __shfl(row[k], j, 16);     // a(j, k) before scaling
#pragma unroll
        for (int j = 0; j <= k; ++j) mkj[j] = __shfl(inv[j], k, 16);        // row k of the inverse so far
        const double lik = lr > k ? row[k] / dk : 0.0;
#pragma unroll
        for (int j = k + 1; j < W; ++j)
          if (lr >= j) row[j] = __builtin_fma(-lik, ajk[j], row[j]);
#pragma unroll
        for (int j = 0; j <= k; ++j) inv[j] = __builtin_fma(-lik, mkj[j], inv[j]);
        if (lr > k) row[k] = lik;
      }
      if (lq == 0) {
#pragma unroll
        for (int c = 0; c < W; ++c) {
          if (lr < w && c <= lr) Dg[lr * (W + 1) + c] = row[c];
          Mi[lr * W + c] = inv[c];
        }
        invd[lr] = lr < w ? 1.0 / row[lr] : 0.0;
      }
    }
    __syncthreads();
    IPM_TICK(2);
    // panel rows: Y^T = L11^-1 A^T (4 products with the accumulator as B operand), L21^T = D^-1 Y^T
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      if ((wv + 4 * i) * 16 >= rows) continue;
      d4 y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int g = 0; g < 4; ++g) y = __builtin_amdgcn_mfma_f64_16x16x4f64(Mi[lr * W + 4 * g + lq], acc[i][g], y, 0, 0, 0);
      const int q = (wv + 4 * i) * 16 + lr, r = rrow[i];
      if (r < 0 || q < w) continue;
      const bool border = r >= G.Nb;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int c = lq + 4 * g;
        const double l = y[g] * invd[c];
        if (c < w && (border || r - (J0 + c) <= G.b)) K[G.at(r, J0 + c)] = l;
        if (border) { BL[(r - G.Nb) * W + c] = l; BY[(r - G.Nb) * W + c] = c < w ? y[g] : 0.0; }
      }
    }
    {                           // the diagonal block is stored as d on the diagonal and L11^-1 below it (what the solves use)
      const int di = t / W, dj = t % W;
      if (di < w && dj <= di) K[G.at(J0 + di, J0 + dj)] = di == dj ? Dg[di * (W + 1) + dj] : Mi[di * W + dj];
    }
    if (t < w) {
      const double dk = Dg[t * (W + 1) + t];
      if (dk > 0) ++npos; else if (dk < 0) ++nneg; else ++nbad;
      if (!(fabs(dk) < 1e300)) ++nbad;
    }
    __syncthreads();
    IPM_TICK(3);
    for (int idx = t; idx < nb * nb; idx += nt) {          // corner -= L_border D L_border^T of this block column
      const int r = idx / nb, c2 = idx % nb;
      if (r < c2) continue;
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < W; ++c) s = __builtin_fma(BL[r * W + c], BY[c2 * W + c], s);
      C[idx] -= s;
    }
    __syncthreads();
    IPM_TICK(4);
    J0 = J1;
  }
  for (int k = 0; k < nb; ++k) {                            // the corner, unblocked, in LDS
    __syncthreads();
    const double dk = C[k * nb + k];
    for (int idx = t; idx < nb * nb; idx += nt) {
      const int r = idx / nb, c = idx % nb;
      if (r > k && c > k && c <= r) C[idx] -= C[r * nb + k] / dk * C[c * nb + k];
    }
    __syncthreads();
    for (int r = k + 1 + t; r < nb; r += nt) C[r * nb + k] /= dk;
  }
  __syncthreads();
  for (int idx = t; idx < nb; idx += nt) {                  // the corner's 16 x 16 diagonal sub-blocks: L -> L^-1, column by column
    const int c0 = idx / W * W, cj = idx, w2 = min(W, nb - c0);
    double x[W];
#pragma unroll
    for (int i = 0; i < W; ++i) {
      double v = c0 + i == cj ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < i; ++k)
        if (i < w2 && c0 + k >= cj) v = __builtin_fma(-C[(c0 + i) * nb + c0 + k], x[k], v);
      x[i] = c0 + i < cj ? 0.0 : v;
    }
    // in place: the 16 columns of a sub-block belong to 16 consecutive lanes of one wave, which has read all of them
    // (the loop above) before any lane writes its column back
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < W; ++i)
      if (i < w2 && c0 + i > cj) C[(c0 + i) * nb + cj] = x[i];
  }
  __syncthreads();
  for (int idx = t; idx < nb * nb; idx += nt) {
    const int r = idx / nb, c = idx % nb;
    if (r >= c) K[G.at(G.Nb + r, G.Nb + c)] = C[idx];
    if (r == c) {
      const double dk = C[idx];
      if (dk > 0) ++npos; else if (dk < 0) ++nneg; else ++nbad;
      if (!(fabs(dk) < 1e300)) ++nbad;
    }
  }
  if (npos) atomicAdd(&cnt[0], npos);
  if (nneg) atomicAdd(&cnt[1], nneg);
  if (nbad) atomicAdd(&cnt[2], nbad);
  __syncthreads();
  IPM_TICK(5);
  if (t == 0) {
    S.npos = cnt[0]; S.nneg = cnt[1]; S.nbad = cnt[2];
#ifdef IPM_TIMING
    for (int i = 0; i < 6; ++i) S.dbg[i] = tc[i];
#endif
  }
}

// L y = r, then x = L^-T D^-1 y, in place in rhs: one workgroup per instance, IPM_W columns per step.  The diagonal
// blocks hold L11^-1, so a step's own 16 unknowns are 16 parallel dot products.  The right-hand side lives in LDS when it
// fits (RL); the diagonal block and each thread's panel row of the NEXT step are fetched while the current one is worked.
template <bool RL>
__global__ __launch_bounds__(256) void kkt_solve_kernel(const double* Kall, long long kstride, KktGeom G, const IpmInst* inst,
                                                        double* rhs_all, int check_status) {
  constexpr int W = IPM_W;
  const int bi = blockIdx.x, t = threadIdx.x, nt = blockDim.x;
  if (check_status && inst[bi].status != 0) return;
  const double* K = Kall + size_t(bi) * kstride;
  double* rg = rhs_all + size_t(bi) * G.Nt;
  extern __shared__ double rsh[];
  double* r = RL ? rsh : rg;
  __shared__ double Dg[W * (W + 1)], ys[W], zs[W], red[4][W];     // Dg: d on the diagonal, L11^-1 below it
  const int di = t / W, dj = t % W;
  const int nbb = (G.Nb + W - 1) / W, ncb = (G.nb + W - 1) / W, nblk = nbb + ncb;
  if (RL) {
    for (int i = t; i < G.Nt; i += nt) rsh[i] = rg[i];
    __syncthreads();
  }
  struct Blk { int J0, J1, nrb, nr, w; };
  auto blk_of = [&](int blk) {
    Blk B;
    B.J0 = blk < nbb ? blk * W : G.Nb + (blk - nbb) * W;
    block_range(G, B.J0, &B.J1, &B.nrb, &B.nr);
    B.w = B.J1 - B.J0;
    return B;
  };
  // this thread's share of a step: one entry of the diagonal block and the 16 factor entries of panel row q = t
  auto fetch = [&](int blk, double& dg, double (&l)[W]) {
    if (blk < 0 || blk >= nblk) return;
    const Blk B = blk_of(blk);
    dg = (di < B.w && dj <= di) ? K[G.at(B.J0 + di, B.J0 + dj)] : 0.0;
    const int row = t < B.nr ? panel_row(G, B.J0, B.J1, B.nrb, t) : -1;
#pragma unroll
    for (int c = 0; c < W; ++c)
      l[c] = (row >= 0 && c < B.w && (row >= G.Nb || row - (B.J0 + c) <= G.b)) ? K[G.at(row, B.J0 + c)] : 0.0;
  };
  double dg, l[W];
  fetch(0, dg, l);
  for (int blk = 0; blk < nblk; ++blk) {
    const Blk B = blk_of(blk);
    if (di < W && dj <= di) Dg[di * (W + 1) + dj] = dg;
    if (t < W) zs[t] = t < B.w ? r[B.J0 + t] : 0.0;
    __syncthreads();
    if (t < W) {                // y = L11^-1 r: 16 lanes, one row each
      double y = zs[t];
#pragma unroll
      for (int k = 0; k < W; ++k)
        if (k < t && t < B.w) y = __builtin_fma(Dg[t * (W + 1) + k], zs[k], y);
      ys[t] = y;
      if (t < B.w) r[B.J0 + t] = y;
    }
    __syncthreads();
    for (int q = t; q < B.nr; q += nt) {
      const int row = panel_row(G, B.J0, B.J1, B.nrb, q);
      double acc = 0.0;
#pragma unroll
      for (int c = 0; c < W; ++c) {
        const double lv = q == t ? l[c] : ((c < B.w && (row >= G.Nb || row - (B.J0 + c) <= G.b)) ? K[G.at(row, B.J0 + c)] : 0.0);
        acc = __builtin_fma(lv, ys[c], acc);
      }
      r[row] -= acc;
    }
    fetch(blk + 1, dg, l);      // in flight across the barrier and the next step's diagonal solve
    __syncthreads();
  }
  fetch(nblk - 1, dg, l);
  for (int blk = nblk - 1; blk >= 0; --blk) {
    const Blk B = blk_of(blk);
    if (di < W && dj <= di) Dg[di * (W + 1) + dj] = dg;
    double p[W];
#pragma unroll
    for (int c = 0; c < W; ++c) p[c] = 0.0;
    for (int q = t; q < B.nr; q += nt) {
      const int row = panel_row(G, B.J0, B.J1, B.nrb, q);
      const double xr = r[row];
#pragma unroll
      for (int c = 0; c < W; ++c) {
        const double lv = q == t ? l[c] : ((c < B.w && (row >= G.Nb || row - (B.J0 + c) <= G.b)) ? K[G.at(row, B.J0 + c)] : 0.0);
        p[c] = __builtin_fma(lv, xr, p[c]);
      }
    }
    fetch(blk - 1, dg, l);      // in flight across the reduction and the diagonal solve
#pragma unroll
    for (int c = 0; c < W; ++c) {
      double sacc = p[c];
      for (int o = 32; o; o >>= 1) sacc += __shfl_down(sacc, o);
      if ((t & 63) == 0) red[t >> 6][c] = sacc;
    }
    __syncthreads();
    if (t < W) zs[t] = t < B.w ? r[B.J0 + t] / Dg[t * (W + 1) + t] - (red[0][t] + red[1][t] + red[2][t] + red[3][t]) : 0.0;
    __syncthreads();
    if (t < B.w) {              // x = L11^-T z
      double x = zs[t];
#pragma unroll
      for (int k = 0; k < W; ++k)
        if (k > t && k < B.w) x = __builtin_fma(Dg[k * (W + 1) + t], zs[k], x);
      r[B.J0 + t] = x;
    }
    __syncthreads();
  }
  if (RL)
    for (int i = t; i < G.Nt; i += nt) rg[i] = rsh[i];
}

// ------------------------------------------------------------------------------------------------ inertia correction
// Algorithm IC: the factorisation is accepted when D has exactly nv positive entries (and no zero / NaN pivot)
__global__ void ipm_inertia_kernel(IpmDev D) {
  const int bi = blockIdx.x * blockDim.x + threadIdx.x;
  if (bi >= D.B) return;
  IpmInst& S = D.inst[bi];
  if (S.status != 0 || !S.refactor) return;
  const IpmOpts& o = D.o;
  if (S.npos == D.nv && S.nbad == 0) {
    S.refactor = 0;
    if (S.delta_w > 0) S.delta_w_last = S.delta_w;
    return;
  }
  if (S.delta_w == 0.0) S.delta_w = S.delta_w_last == 0.0 ? o.delta_w_first : fmax(o.delta_w_min, o.kw_dec * S.delta_w_last);
  else S.delta_w *= S.delta_w_last == 0.0 ? o.kw_inc_first : o.kw_inc;
  if (S.delta_w > o.delta_w_max) { S.status = 4; return; }
  atomicAdd(&D.cnt[1], 1);
}

// ------------------------------------------------------------------------------------------------ direction
__global__ __launch_bounds__(256) void ipm_direction_kernel(IpmDev D) {
  __shared__ double sh[4];
  const int bi = blockIdx.x, t = threadIdx.x;
  IpmInst& S = D.inst[bi];
  if (S.status != 0) return;
  const size_t o = size_t(bi) * D.nv;
  const double* sol = D.rhs + size_t(bi) * D.Nt;
  double amax = 1.0, az = 1.0, dphi = 0.0, bad = 0.0;
  const double mu = S.mu, tau = S.tau;
  if (S.mode == 1) {            // Gauss-Newton step of the restoration: slope = g_b^T d + (A d)^T c, A d = w - c
    double slope = 0.0;
    for (int i = t; i < D.nv; i += blockDim.x) {
      const double l = D.vl[o + i], u = D.vu[o + i], vi = D.v[o + i];
      double d = 0.0;
      if (l != u) {
        d = sol[D.pos[i]];
        if (!(fabs(d) < 1e300)) bad = 1;
        double gb = S.zeta * D.dr2[o + i] * (vi - D.vR[o + i]);
        if (l > -IPM_INF) { const double s = vi - l; if (d < 0) amax = fmin(amax, -tau * s / d); gb -= mu / s; }
        if (u < IPM_INF) { const double s = u - vi; if (d > 0) amax = fmin(amax, tau * s / d); gb += mu / s; }
        slope += gb * d;
      }
      D.dv[o + i] = d;
      D.dzL[o + i] = 0.0;
      D.dzU[o + i] = 0.0;
    }
    for (int r = t; r < D.m; r += blockDim.x) {
      const double cr = D.c[size_t(bi) * D.m + r];
      slope += (sol[D.pos[D.nv + r]] - cr) * cr;
      D.dlam[size_t(bi) * D.m + r] = 0.0;
    }
    amax = block_red(amax, 2, sh); slope = block_red(slope, 0, sh); bad = block_red(bad, 1, sh);
    if (t != 0) return;
    if (bad != 0) { S.status = 5; return; }
    S.alpha_max = amax; S.alpha_z = 0.0; S.alpha = amax; S.slope = slope; S.dphi = slope;
    S.ls = 0; S.accepted = 0; S.armijo = 0;
    atomicAdd(&D.cnt[2], 1);
    return;
  }
  for (int i = t; i < D.nv; i += blockDim.x) {
    const double l = D.vl[o + i], u = D.vu[o + i], vi = D.v[o + i];
    double d = 0.0, dl = 0.0, du = 0.0;
    if (l != u) {
      d = sol[D.pos[i]];
      if (!(fabs(d) < 1e300)) bad = 1;
      double gphi = i < D.n ? D.grad[size_t(bi) * D.n + i] : 0.0;
      if (l > -IPM_INF) {
        const double s = vi - l, z = D.zL[o + i];
        dl = mu / s - z - z / s * d;                       // (12)
        if (d < 0) amax = fmin(amax, -tau * s / d);        // (15a)
        if (dl < 0) az = fmin(az, -tau * z / dl);          // (15b)
        gphi -= mu / s;
      }
      if (u < IPM_INF) {
        const double s = u - vi, z = D.zU[o + i];
        du = mu / s - z + z / s * d;
        if (d > 0) amax = fmin(amax, tau * s / d);
        if (du < 0) az = fmin(az, -tau * z / du);
        gphi += mu / s;
      }
      dphi += gphi * d;
    }
    D.dv[o + i] = d;
    D.dzL[o + i] = dl;
    D.dzU[o + i] = du;
  }
  for (int r = t; r < D.m; r += blockDim.x) D.dlam[size_t(bi) * D.m + r] = sol[D.pos[D.nv + r]];
  amax = block_red(amax, 2, sh); az = block_red(az, 2, sh); dphi = block_red(dphi, 0, sh); bad = block_red(bad, 1, sh);
  if (t != 0) return;
  if (bad != 0) { S.status = 5; return; }
  const IpmOpts& op = D.o;
  S.alpha_max = amax; S.alpha_z = az; S.alpha = amax; S.dphi = dphi;
  double amin = op.gamma_theta;                            // (23)
  if (dphi < 0) {
    amin = fmin(amin, op.gamma_phi * S.theta / (-dphi));
    if (S.theta <= S.theta_min) amin = fmin(amin, op.delta * pow(S.theta, op.s_theta) / pow(-dphi, op.s_phi));
  }
  S.alpha_min = op.gamma_alpha * amin;
  S.ls = 0; S.accepted = 0; S.armijo = 0;
  atomicAdd(&D.cnt[2], 1);
}

// ------------------------------------------------------------------------------------------------ line search
__global__ void ipm_trial_kernel(IpmDev D) {
  const int bi = blockIdx.y;
  const IpmInst& S = D.inst[bi];
  if (S.status != 0 || S.accepted) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < D.n) D.xt[size_t(bi) * D.n + i] = D.v[size_t(bi) * D.nv + i] + S.alpha * D.dv[size_t(bi) * D.nv + i];
}
__global__ __launch_bounds__(256) void ipm_accept_kernel(IpmDev D) {
  __shared__ double sh[4];
  const int bi = blockIdx.x, t = threadIdx.x;
  IpmInst& S = D.inst[bi];
  if (S.status != 0 || S.accepted) return;
  const size_t o = size_t(bi) * D.nv;
  const double a = S.alpha;
  double th = 0.0, ln = 0.0, bad = 0.0, csq = 0.0, qd = 0.0;
  const bool resto = S.mode == 1;
  for (int i = t; i < D.nv; i += blockDim.x) {
    const double l = D.vl[o + i], u = D.vu[o + i];
    if (l == u) continue;
    const double vt = D.v[o + i] + a * D.dv[o + i];
    if (l > -IPM_INF) ln += log(vt - l);
    if (u < IPM_INF) ln += log(u - vt);
    if (resto) { const double dd = vt - D.vR[o + i]; qd += D.dr2[o + i] * dd * dd; }
  }
  for (int r = t; r < D.m; r += blockDim.x) {
    const int s = D.row_slack[r];
    const double gr = D.gt[size_t(bi) * D.sg + r];
    const double cr = s < 0 ? gr - D.gl[r] : gr - (D.v[o + D.n + s] + a * D.dv[o + D.n + s]);
    if (!(fabs(cr) < 1e300)) bad = 1;
    th += fabs(cr);
    csq += cr * cr;
  }
  th = block_red(th, 0, sh); ln = block_red(ln, 0, sh); bad = block_red(bad, 1, sh);
  if (resto) { csq = block_red(csq, 0, sh); qd = block_red(qd, 0, sh); }
  if (t != 0) return;
  const IpmOpts& op = D.o;
  if (resto) {                  // Armijo on psi = 1/2 |c|^2 + zeta/2 |D_R (v - v_R)|^2 - mu sum ln
    const double psit = 0.5 * csq + 0.5 * S.zeta * qd - S.mu * ln;
    if (bad == 0 && fabs(psit) < 1e300 && psit <= S.psi + 1e-4 * a * S.slope) { S.accepted = 1; return; }
    S.alpha = 0.5 * a;
    S.ls += 1;
    if (S.ls >= op.max_ls) { S.status = 3; return; }
    atomicAdd(&D.cnt[2], 1);
    return;
  }
  const double ft = D.objt[bi];
  if (!(fabs(ft) < 1e300) || !(fabs(ln) < 1e300)) bad = 1;
  const double phit = ft - S.mu * ln;
  const double slack = 10.0 * 2.220446049250313e-16 * fabs(S.phi);     // Ipopt's rounding allowance in the phi comparisons
  bool ok = false;
  if (bad == 0 && th <= S.theta_max) {
    bool dominated = false;
    const double* F = D.filt + size_t(bi) * 2 * IPM_FMAX;
    for (int k = 0; k < S.nfilt; ++k)
      if (th >= F[2 * k] && phit >= F[2 * k + 1]) dominated = true;
    if (!dominated) {
      const bool sw = S.dphi < 0 && a * pow(-S.dphi, op.s_phi) > op.delta * pow(S.theta, op.s_theta);   // (19)
      if (S.theta <= S.theta_min && sw) {
        ok = phit - S.phi - op.eta_phi * a * S.dphi <= slack;                                             // (20)
        if (ok) S.armijo = 1;
      } else {
        ok = th <= (1.0 - op.gamma_theta) * S.theta || phit - (S.phi - op.gamma_phi * S.theta) <= slack;  // (18)
      }
    }
  }
  if (ok) { S.accepted = 1; return; }
  S.alpha = 0.5 * a;
  S.ls += 1;
  if (S.alpha < S.alpha_min || S.ls > op.max_ls) {
    if (S.err0 <= op.acceptable_tol) S.status = 6;             // nothing left to gain: Ipopt reports the acceptable level here too
    else if (op.resto && S.theta > op.tol) S.enter_resto = 1;  // Ipopt switches to its restoration phase here
    else S.status = 3;
    return;
  }
  atomicAdd(&D.cnt[2], 1);
}

// ------------------------------------------------------------------------------------------------ step
__global__ __launch_bounds__(256) void ipm_update_kernel(IpmDev D) {
  const int bi = blockIdx.x, t = threadIdx.x;
  IpmInst& S = D.inst[bi];
  if (S.status != 0) return;
  const size_t o = size_t(bi) * D.nv;
  if (S.enter_resto) {          // the line search gave up at an infeasible point: start the feasibility restoration from it
    for (int i = t; i < D.nv; i += blockDim.x) {
      const double vi = D.v[o + i], sc = fmax(1.0, fabs(vi));
      D.vR[o + i] = vi;
      D.dr2[o + i] = 1.0 / (sc * sc);
    }
    if (t == 0) {
      if (S.nfilt < IPM_FMAX) {
        double* F = D.filt + size_t(bi) * 2 * IPM_FMAX;
        F[2 * S.nfilt] = (1.0 - D.o.gamma_theta) * S.theta;
        F[2 * S.nfilt + 1] = S.phi - D.o.gamma_phi * S.theta;
        S.nfilt += 1;
      }
      S.th0 = S.theta; S.zeta = sqrt(S.mu); S.mode = 1; S.resto_it = 0; S.enter_resto = 0;
    }
    return;
  }
  if (!S.accepted) return;
  if (S.mode == 1) {
    for (int i = t; i < D.nv; i += blockDim.x)
      if (D.vl[o + i] != D.vu[o + i]) D.v[o + i] += S.alpha * D.dv[o + i];
    if (t == 0) {
      if (D.trace && S.iter < D.trace_cap) {
        double* R = D.trace + (size_t(bi) * D.trace_cap + S.iter) * IPM_TRACE;
        R[0] = S.f; R[1] = S.theta; R[2] = S.mu; R[3] = S.alpha; R[4] = 0.0; R[5] = 0.0; R[6] = S.err0; R[7] = -1.0;
      }
      S.resto_it += 1;
      S.iter += 1;
    }
    return;
  }
  const double a = S.alpha, az = S.alpha_z, mu = S.mu, ks = D.o.kappa_sigma;
  for (int i = t; i < D.nv; i += blockDim.x) {
    const double l = D.vl[o + i], u = D.vu[o + i];
    if (l == u) continue;
    const double vi = D.v[o + i] + a * D.dv[o + i];
    D.v[o + i] = vi;
    if (l > -IPM_INF) {
      const double s = vi - l;
      D.zL[o + i] = fmax(fmin(D.zL[o + i] + az * D.dzL[o + i], ks * mu / s), mu / (ks * s));   // (16)
    }
    if (u < IPM_INF) {
      const double s = u - vi;
      D.zU[o + i] = fmax(fmin(D.zU[o + i] + az * D.dzU[o + i], ks * mu / s), mu / (ks * s));
    }
  }
  for (int r = t; r < D.m; r += blockDim.x) D.lam[size_t(bi) * D.m + r] += a * D.dlam[size_t(bi) * D.m + r];
  if (t == 0) {
    if (!S.armijo && S.nfilt < IPM_FMAX) {       // (22)
      double* F = D.filt + size_t(bi) * 2 * IPM_FMAX;
      F[2 * S.nfilt] = (1.0 - D.o.gamma_theta) * S.theta;
      F[2 * S.nfilt + 1] = S.phi - D.o.gamma_phi * S.theta;
      S.nfilt += 1;
    }
    if (D.trace && S.iter < D.trace_cap) {
      double* R = D.trace + (size_t(bi) * D.trace_cap + S.iter) * IPM_TRACE;
      R[0] = S.f; R[1] = S.theta; R[2] = S.mu; R[3] = S.alpha; R[4] = S.alpha_z; R[5] = S.delta_w; R[6] = S.err0; R[7] = double(S.ls);
    }
    S.iter += 1;
  }
}

}  // namespace rpm

// =================================================================================================== host side + ABI
using namespace rpm;

struct rpm_engine { rpm::Engine e; };

struct rpm_ipm {
  rpm_engine* eng = nullptr;
  IpmPlan plan;
  IpmDev D{};
  std::vector<void*> allocs;
  int* h_cnt = nullptr;           // page-locked mirror of D.cnt
  size_t factor_lds = 0;
  int factor_mt = IPM_MT;
  std::string err;
  std::vector<IpmInst> h_inst;
  int total_factorizations = 0, total_iterations = 0, total_trials = 0;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // around the factorisation and the substitution of an iteration
  double factor_ms = 0.0, solve_ms = 0.0;
  bool solve_pending = false;
  ~rpm_ipm() {
    for (void* p : allocs) (void)hipFree(p);
    if (h_cnt) (void)hipHostFree(h_cnt);
    for (hipEvent_t e2 : ev)
      if (e2) (void)hipEventDestroy(e2);
  }
};

#define IPM_TRY(h, call)                                                      \
  do {                                                                        \
    hipError_t _s = (call);                                                   \
    if (_s != hipSuccess) {                                                   \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(_s);          \
      return RPM_E_DEVICE;                                                    \
    }                                                                         \
  } while (0)

namespace {
template <class T>
int ipm_alloc(rpm_ipm* h, T** dst, size_t count, const T* src = nullptr) {
  void* p = nullptr;
  IPM_TRY(h, hipMalloc(&p, (count ? count : 1) * sizeof(T)));
  h->allocs.push_back(p);
  *dst = static_cast<T*>(p);
  if (src && count) IPM_TRY(h, hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice));
  return RPM_OK;
}
template <class T>
int ipm_alloc_c(rpm_ipm* h, const T** dst, const std::vector<T>& src) {
  T* p = nullptr;
  int rc = ipm_alloc(h, &p, src.size(), src.data());
  *dst = p;
  return rc;
}
KktGeom geom_of(const IpmPlan& p) { return KktGeom{p.Nt, p.Nb, p.nb, p.b, p.CS}; }

int fetch_counts(rpm_ipm* h, hipStream_t st) {
  IPM_TRY(h, hipMemcpyAsync(h->h_cnt, h->D.cnt, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
  IPM_TRY(h, hipStreamSynchronize(st));
  return RPM_OK;
}
int launch_check(rpm_ipm* h, const char* what) {
  hipError_t s = hipGetLastError();
  if (s != hipSuccess) {
    h->err = std::string(what) + ": " + hipGetErrorString(s);
    return RPM_E_DEVICE;
  }
  return RPM_OK;
}
int factor_and_solve_launch(rpm_ipm* h, hipStream_t st, bool factor, bool solve, int check_status) {
  const IpmDev& D = h->D;
  if (factor) {
    if (h->factor_mt == 4)
      hipLaunchKernelGGL(kkt_factor_kernel<4>, dim3(unsigned(D.B)), dim3(256), h->factor_lds, st, D.K, D.kstride, geom_of(h->plan), D.inst);
    else
      hipLaunchKernelGGL(kkt_factor_kernel<IPM_MT>, dim3(unsigned(D.B)), dim3(256), h->factor_lds, st, D.K, D.kstride, geom_of(h->plan),
                         D.inst);
  }
  if (solve) {
    if (size_t(D.Nt) * sizeof(double) <= 48 * 1024)
      hipLaunchKernelGGL(kkt_solve_kernel<true>, dim3(unsigned(D.B)), dim3(256), size_t(D.Nt) * sizeof(double), st, D.K, D.kstride,
                         geom_of(h->plan), D.inst, D.rhs, check_status);
    else
      hipLaunchKernelGGL(kkt_solve_kernel<false>, dim3(unsigned(D.B)), dim3(256), 0, st, D.K, D.kstride, geom_of(h->plan), D.inst, D.rhs,
                         check_status);
  }
  return launch_check(h, "kkt kernels");
}
}  // namespace

extern "C" {

int rpm_ipm_create(rpm_engine* eng, rpm_ipm** out) {
  if (!eng || !out) return RPM_E_INVALID;
  *out = nullptr;
  Engine& e = eng->e;
  if (e.hessian_mode != RPM_HESSIAN_EXACT) {
    e.err = "rpm_ipm_create: the engine must be created with hessian-approximation=exact";
    return RPM_E_UNSUPPORTED;
  }
  if (e.shard_world > 1) {
    e.err = "rpm_ipm_create: interval-sharded engines are not supported (shard instances across ranks instead)";
    return RPM_E_UNSUPPORTED;
  }
  if (!e.dev) {
    int rc = device_init(e, 0);
    if (rc) return rc;
  }
  int rc = ensure_hessian(e);
  if (rc) return rc;
  rpm_ipm* h = new (std::nothrow) rpm_ipm;
  if (!h) return RPM_E_INVALID;
  h->eng = eng;
  std::string why;
  rc = build_ipm_plan(e, h->plan, &why);
  if (rc) {
    e.err = "rpm_ipm_create: " + why;
    delete h;
    return rc;
  }
  const IpmPlan& p = h->plan;
  IpmDev& D = h->D;
  const size_t B = size_t(e.n_instances);
  D.B = int(B); D.n = p.n; D.m = p.m; D.ns = p.ns; D.nv = p.nv; D.Nt = p.Nt; D.Nb = p.Nb; D.nb = p.nb; D.b = p.b; D.CS = p.CS;
  D.nnz_jac = e.nnz_jac; D.nnz_h = e.nnz_h;
  D.sg = e.stride_g(); D.sv = e.stride_values(); D.kstride = p.storage();
  auto fail = [&](int code) { e.err = "rpm_ipm_create: " + h->err; delete h; return code; };
#define A_(call) do { int _r = (call); if (_r) return fail(_r); } while (0)
  A_(ipm_alloc_c(h, &D.pos, p.pos)); A_(ipm_alloc_c(h, &D.row_slack, p.row_slack)); A_(ipm_alloc_c(h, &D.slack_row, p.slack_row));
  A_(ipm_alloc_c(h, &D.jac_dst, p.jac_dst)); A_(ipm_alloc_c(h, &D.hes_dst, p.hes_dst));
  A_(ipm_alloc_c(h, &D.diag_dst, p.diag_dst)); A_(ipm_alloc_c(h, &D.slk_dst, p.slk_dst)); A_(ipm_alloc_c(h, &D.jt_ptr, p.jt_ptr));
  A_(ipm_alloc_c(h, &D.jt_ent, p.jt_ent)); A_(ipm_alloc_c(h, &D.jt_row, p.jt_row));
  A_(ipm_alloc_c(h, &D.gl, e.gl)); A_(ipm_alloc_c(h, &D.gu, e.gu));
  A_(ipm_alloc(h, &D.v, B * p.nv)); A_(ipm_alloc(h, &D.vl, B * p.nv)); A_(ipm_alloc(h, &D.vu, B * p.nv));
  A_(ipm_alloc(h, &D.zL, B * p.nv)); A_(ipm_alloc(h, &D.zU, B * p.nv)); A_(ipm_alloc(h, &D.lam, B * p.m));
  A_(ipm_alloc(h, &D.dv, B * p.nv)); A_(ipm_alloc(h, &D.dlam, B * p.m)); A_(ipm_alloc(h, &D.dzL, B * p.nv));
  A_(ipm_alloc(h, &D.dzU, B * p.nv)); A_(ipm_alloc(h, &D.glag, B * p.nv)); A_(ipm_alloc(h, &D.c, B * p.m));
  A_(ipm_alloc(h, &D.rhs, B * p.Nt)); A_(ipm_alloc(h, &D.K, B * size_t(p.storage()))); A_(ipm_alloc(h, &D.filt, B * 2 * IPM_FMAX));
  A_(ipm_alloc(h, &D.xe, B * p.n)); A_(ipm_alloc(h, &D.xt, B * p.n)); A_(ipm_alloc(h, &D.grad, B * p.n));
  A_(ipm_alloc(h, &D.g, B * size_t(D.sg))); A_(ipm_alloc(h, &D.jac, B * size_t(D.sv))); A_(ipm_alloc(h, &D.hess, B * size_t(e.nnz_h)));
  A_(ipm_alloc(h, &D.obj, B)); A_(ipm_alloc(h, &D.gt, B * size_t(D.sg))); A_(ipm_alloc(h, &D.objt, B));
  A_(ipm_alloc(h, &D.inst, B)); A_(ipm_alloc(h, &D.cnt, size_t(4)));
  A_(ipm_alloc(h, &D.vR, B * p.nv)); A_(ipm_alloc(h, &D.dr2, B * p.nv));
#undef A_
  if (hipHostMalloc(reinterpret_cast<void**>(&h->h_cnt), 4 * sizeof(int)) != hipSuccess) { h->err = "hipHostMalloc"; return fail(RPM_E_DEVICE); }
  // variable bounds of every instance default to the engine's
  {
    std::vector<double> l(B * p.nv, 0.0), u(B * p.nv, 0.0);
    for (size_t bi = 0; bi < B; ++bi)
      for (int i = 0; i < p.n; ++i) { l[bi * p.nv + i] = e.xl[i]; u[bi * p.nv + i] = e.xu[i]; }
    if (hipMemcpy(D.vl, l.data(), l.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(D.vu, u.data(), u.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { h->err = "hipMemcpy"; return fail(RPM_E_DEVICE); }
  }
  h->factor_mt = IPM_W + p.b + p.nb <= 256 ? 4 : IPM_MT;
  if (IPM_W + p.b + p.nb > 4 * IPM_MT * 16) {
    h->err = "band + border of " + std::to_string(p.b + p.nb) + " rows exceeds the factorisation's 512 rows per block column";
    return fail(RPM_E_UNSUPPORTED);
  }
  h->factor_lds = (size_t(p.b + 8) * IPM_W + size_t(IPM_W) * (IPM_W + 1) + size_t(IPM_W) * IPM_W + IPM_W + 2 * size_t(p.nb) * IPM_W +
                   size_t(p.nb) * p.nb) * sizeof(double);
  if (h->factor_lds > 150 * 1024) {
    h->err = "band of " + std::to_string(p.b) + " and border of " + std::to_string(p.nb) + " rows do not fit the factorisation's LDS";
    return fail(RPM_E_UNSUPPORTED);
  }
  if (h->factor_lds > 48 * 1024 &&
      hipFuncSetAttribute(h->factor_mt == 4 ? reinterpret_cast<const void*>(kkt_factor_kernel<4>) : reinterpret_cast<const void*>(kkt_factor_kernel<IPM_MT>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, int(h->factor_lds)) != hipSuccess) { h->err = "hipFuncSetAttribute"; return fail(RPM_E_DEVICE); }
  h->h_inst.resize(B);
  for (hipEvent_t& e2 : h->ev)
    if (hipEventCreate(&e2) != hipSuccess) { h->err = "hipEventCreate"; return fail(RPM_E_DEVICE); }
  *out = h;
  return RPM_OK;
}

void rpm_ipm_destroy(rpm_ipm* h) { delete h; }
const char* rpm_ipm_last_error(const rpm_ipm* h) { return h ? h->err.c_str() : "null solver"; }

int rpm_ipm_set_option(rpm_ipm* h, const char* key, double value) {
  if (!h || !key) return RPM_E_INVALID;
  IpmOpts& o = h->D.o;
  const std::string k(key);
  if (k == "tol") o.tol = value;
  else if (k == "max_iter") o.max_iter = int(value);
  else if (k == "mu_init") o.mu_init = value;
  else if (k == "bound_push") o.bound_push = value;
  else if (k == "bound_frac") o.bound_frac = value;
  else if (k == "delta_c") o.delta_c = value;
  else if (k == "max_line_search") o.max_ls = int(value);
  else if (k == "restoration") o.resto = value != 0.0;
  else if (k == "acceptable_tol") o.acceptable_tol = value;
  else if (k == "acceptable_iter") o.acceptable_iter = int(value);
  else if (k == "restoration_max_iter") o.resto_max = int(value);
  else if (k == "trace") {          // keep the first `value` iterations of every instance (rpm_ipm_get_trace)
    const int cap = int(value);
    if (cap < 0 || cap > 100000) { h->err = "trace: 0 ... 100000 records"; return RPM_E_INVALID; }
    h->D.trace = nullptr;
    h->D.trace_cap = 0;
    if (cap > 0) {
      int rc = ipm_alloc(h, &h->D.trace, size_t(h->D.B) * cap * IPM_TRACE);
      if (rc) return rc;
      h->D.trace_cap = cap;
    }
  }
  else { h->err = "unknown option " + k; return RPM_E_INVALID; }
  return RPM_OK;
}

int rpm_ipm_get_info(rpm_ipm* h, int* kkt_order, int* band_order, int* half_bandwidth, int* border, long long* storage_doubles,
                     int* n_slacks) {
  if (!h) return RPM_E_INVALID;
  if (kkt_order) *kkt_order = h->plan.Nt;
  if (band_order) *band_order = h->plan.Nb;
  if (half_bandwidth) *half_bandwidth = h->plan.b;
  if (border) *border = h->plan.nb;
  if (storage_doubles) *storage_doubles = h->plan.storage();
  if (n_slacks) *n_slacks = h->plan.ns;
  return RPM_OK;
}

int rpm_ipm_get_stats(rpm_ipm* h, int* iterations, int* factorizations, int* trial_points) {
  if (!h) return RPM_E_INVALID;
  if (iterations) *iterations = h->total_iterations;
  if (factorizations) *factorizations = h->total_factorizations;
  if (trial_points) *trial_points = h->total_trials;
  return RPM_OK;
}

int rpm_ipm_get_kernel_times(rpm_ipm* h, double* factor_ms, double* substitution_ms) {
  if (!h) return RPM_E_INVALID;
  if (factor_ms) *factor_ms = h->factor_ms;
  if (substitution_ms) *substitution_ms = h->solve_ms;
  return RPM_OK;
}

int rpm_ipm_get_restorations(rpm_ipm* h, int* per_instance) {
  if (!h || !per_instance || h->h_inst.empty()) return RPM_E_INVALID;
  for (int bi = 0; bi < h->D.B; ++bi) per_instance[bi] = h->h_inst[bi].n_resto;
  return RPM_OK;
}

int rpm_ipm_get_trace(rpm_ipm* h, int instance, int capacity, double* records, int* n_records) {
  if (!h || instance < 0 || instance >= h->D.B || !records || !n_records) return RPM_E_INVALID;
  if (!h->D.trace || h->h_inst.empty()) { *n_records = 0; return RPM_OK; }
  const int n = std::min(std::min(h->h_inst[instance].iter, h->D.trace_cap), capacity);
  IPM_TRY(h, hipMemcpy(records, h->D.trace + size_t(instance) * h->D.trace_cap * IPM_TRACE, size_t(n) * IPM_TRACE * sizeof(double),
                       hipMemcpyDeviceToHost));
  *n_records = n;
  return RPM_OK;
}

int rpm_ipm_set_bounds(rpm_ipm* h, int instance, const double* x_l, const double* x_u) {
  if (!h || !x_l || !x_u || instance < 0 || instance >= h->D.B) return RPM_E_INVALID;
  const IpmPlan& p = h->plan;
  for (int i = 0; i < p.n; ++i)
    if ((x_l[i] == x_u[i]) != (p.fixed[i] != 0)) {
      h->err = "rpm_ipm_set_bounds: variable " + std::to_string(i) + " changes between fixed and free (the KKT layout is shared by all instances)";
      return RPM_E_INVALID;
    }
  IPM_TRY(h, hipMemcpy(h->D.vl + size_t(instance) * p.nv, x_l, p.n * sizeof(double), hipMemcpyHostToDevice));
  IPM_TRY(h, hipMemcpy(h->D.vu + size_t(instance) * p.nv, x_u, p.n * sizeof(double), hipMemcpyHostToDevice));
  return RPM_OK;
}

/* variable bounds of all instances at once: x_l, x_u are n_instances x n (host), e.g. the measured initial states of a
 * receding-horizon sweep; two copies instead of 2 n_instances */
int rpm_ipm_set_all_bounds(rpm_ipm* h, const double* x_l, const double* x_u) {
  if (!h || !x_l || !x_u) return RPM_E_INVALID;
  const IpmPlan& p = h->plan;
  const size_t B = size_t(h->D.B);
  for (size_t bi = 0; bi < B; ++bi)
    for (int i = 0; i < p.n; ++i)
      if ((x_l[bi * p.n + i] == x_u[bi * p.n + i]) != (p.fixed[i] != 0)) {
        h->err = "rpm_ipm_set_all_bounds: instance " + std::to_string(bi) + ", variable " + std::to_string(i) +
                 " changes between fixed and free (the KKT layout is shared by all instances)";
        return RPM_E_INVALID;
      }
  IPM_TRY(h, hipMemcpy2D(h->D.vl, size_t(p.nv) * sizeof(double), x_l, size_t(p.n) * sizeof(double), size_t(p.n) * sizeof(double), B,
                         hipMemcpyHostToDevice));
  IPM_TRY(h, hipMemcpy2D(h->D.vu, size_t(p.nv) * sizeof(double), x_u, size_t(p.n) * sizeof(double), size_t(p.n) * sizeof(double), B,
                         hipMemcpyHostToDevice));
  return RPM_OK;
}

/* test hook: factor + solve the caller's matrices (B x storage doubles in the band + border layout, lower triangle)
 * against B right-hand sides in KKT order; returns the solutions and the signs of D */
int rpm_ipm_debug_solve(rpm_ipm* h, const double* k_storage, const double* rhs, double* sol, int* n_pos, int* n_neg) {
  if (!h || !k_storage || !rhs || !sol) return RPM_E_INVALID;
  const IpmPlan& p = h->plan;
  IpmDev& D = h->D;
  hipStream_t st = static_cast<hipStream_t>(dev_stream(h->eng->e));
  std::vector<IpmInst> inst(D.B);
  for (auto& s : inst) { s = IpmInst{}; s.refactor = 1; }
  IPM_TRY(h, hipMemcpy(D.inst, inst.data(), inst.size() * sizeof(IpmInst), hipMemcpyHostToDevice));
  IPM_TRY(h, hipMemcpy(D.K, k_storage, size_t(D.B) * p.storage() * sizeof(double), hipMemcpyHostToDevice));
  IPM_TRY(h, hipMemcpy(D.rhs, rhs, size_t(D.B) * p.Nt * sizeof(double), hipMemcpyHostToDevice));
  int rc = factor_and_solve_launch(h, st, true, true, 0);
  if (rc) return rc;
  IPM_TRY(h, hipStreamSynchronize(st));
  IPM_TRY(h, hipMemcpy(sol, D.rhs, size_t(D.B) * p.Nt * sizeof(double), hipMemcpyDeviceToHost));
  IPM_TRY(h, hipMemcpy(inst.data(), D.inst, inst.size() * sizeof(IpmInst), hipMemcpyDeviceToHost));
#ifdef IPM_TIMING
  fprintf(stderr, "factor phases of instance 0 [100 MHz ticks]: T %lld  k-loop %lld  diag %lld  panel %lld  corner %lld  tail %lld\n",
          inst[0].dbg[0], inst[0].dbg[1], inst[0].dbg[2], inst[0].dbg[3], inst[0].dbg[4], inst[0].dbg[5]);
#endif
  for (int bi = 0; bi < D.B; ++bi) {
    if (n_pos) n_pos[bi] = inst[bi].npos;
    if (n_neg) n_neg[bi] = inst[bi].nneg;
  }
  return RPM_OK;
}

/* KKT position of every unknown ([0,n) variables, then the slacks, then the m multipliers) — for tests and tools */
int rpm_ipm_get_permutation(rpm_ipm* h, int* pos, int capacity) {
  if (!h || !pos || capacity < h->plan.Nt) return RPM_E_INVALID;
  std::memcpy(pos, h->plan.pos.data(), sizeof(int) * h->plan.Nt);
  return RPM_OK;
}

int rpm_ipm_solve_dev(rpm_ipm* h, double* d_x, double* d_lambda, double* obj, int* status, int* iterations, double* kkt_error) {
  if (!h || !d_x) return RPM_E_INVALID;
  Engine& e = h->eng->e;
  IpmDev& D = h->D;
  const IpmPlan& p = h->plan;
  hipStream_t st = static_cast<hipStream_t>(dev_stream(e));
  const unsigned B = unsigned(D.B);
  const dim3 gx((p.n + 255) / 256, B);
  auto eng_fail = [&](int rc) { h->err = e.err; return rc; };
  h->total_factorizations = h->total_iterations = h->total_trials = 0;
  h->factor_ms = h->solve_ms = 0.0;
  h->solve_pending = false;

  IPM_TRY(h, hipMemcpyAsync(D.xt, d_x, size_t(B) * p.n * sizeof(double), hipMemcpyDeviceToDevice, st));
  hipLaunchKernelGGL(ipm_init_kernel, dim3(B), dim3(256), 0, st, D, d_x);
  hipLaunchKernelGGL(ipm_pack_x_kernel, gx, dim3(256), 0, st, D);
  int rc = dev_eval_cons(e, D.xe, D.g, nullptr, 1 | 4, st);
  if (rc) return eng_fail(rc);
  hipLaunchKernelGGL(ipm_init_slack_kernel, dim3(B), dim3(256), 0, st, D);
  if ((rc = launch_check(h, "ipm_init"))) return rc;

  const int assemble_blocks = std::max(1, std::min(64, (std::max(e.nnz_jac, e.nnz_h) + 255) / 256));
  const int zero_blocks = int(std::max<long long>(1, std::min<long long>(256, p.storage() / 2 / 256 + 1)));
  for (;;) {
    hipLaunchKernelGGL(ipm_pack_x_kernel, gx, dim3(256), 0, st, D);
    if ((rc = dev_eval_obj(e, D.xe, D.obj, D.grad, st))) return eng_fail(rc);
    if ((rc = dev_eval_cons(e, D.xe, D.g, D.jac, 3 | 4, st))) return eng_fail(rc);
    IPM_TRY(h, hipMemsetAsync(D.cnt, 0, 4 * sizeof(int), st));
    hipLaunchKernelGGL(ipm_residual_kernel, dim3(B), dim3(256), 0, st, D);
    if ((rc = fetch_counts(h, st))) return rc;
    if (h->h_cnt[0] == 0) break;
    h->total_iterations += 1;
    if ((rc = dev_eval_h(e, D.xe, 1.0, D.lam, D.hess, st))) return eng_fail(rc);
    for (int tries = 0; tries < 80; ++tries) {
      IPM_TRY(h, hipMemsetAsync(D.cnt + 1, 0, sizeof(int), st));
      hipLaunchKernelGGL(ipm_zero_kernel, dim3(unsigned(zero_blocks), B), dim3(256), 0, st, D);
      hipLaunchKernelGGL(ipm_assemble_kernel, dim3(unsigned(assemble_blocks), B), dim3(256), 0, st, D);
      IPM_TRY(h, hipEventRecord(h->ev[0], st));
      if ((rc = factor_and_solve_launch(h, st, true, false, 1))) return rc;
      IPM_TRY(h, hipEventRecord(h->ev[1], st));
      hipLaunchKernelGGL(ipm_inertia_kernel, dim3((B + 255) / 256), dim3(256), 0, st, D);
      h->total_factorizations += 1;
      if ((rc = fetch_counts(h, st))) return rc;
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, h->ev[0], h->ev[1]) == hipSuccess) h->factor_ms += ms;
      if (h->h_cnt[1] == 0) break;
    }
    IPM_TRY(h, hipEventRecord(h->ev[2], st));
    if ((rc = factor_and_solve_launch(h, st, false, true, 1))) return rc;
    IPM_TRY(h, hipEventRecord(h->ev[3], st));
    h->solve_pending = true;
    IPM_TRY(h, hipMemsetAsync(D.cnt + 2, 0, sizeof(int), st));
    hipLaunchKernelGGL(ipm_direction_kernel, dim3(B), dim3(256), 0, st, D);
    for (int ls = 0; ls <= D.o.max_ls + 1; ++ls) {
      hipLaunchKernelGGL(ipm_trial_kernel, gx, dim3(256), 0, st, D);
      if ((rc = dev_eval_obj(e, D.xt, D.objt, nullptr, st))) return eng_fail(rc);
      if ((rc = dev_eval_cons(e, D.xt, D.gt, nullptr, 1 | 4, st))) return eng_fail(rc);
      IPM_TRY(h, hipMemsetAsync(D.cnt + 2, 0, sizeof(int), st));
      hipLaunchKernelGGL(ipm_accept_kernel, dim3(B), dim3(256), 0, st, D);
      h->total_trials += 1;
      if ((rc = fetch_counts(h, st))) return rc;
      if (h->solve_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, h->ev[2], h->ev[3]) == hipSuccess) h->solve_ms += ms;
        h->solve_pending = false;
      }
      if (h->h_cnt[2] == 0) break;
    }
    hipLaunchKernelGGL(ipm_update_kernel, dim3(B), dim3(256), 0, st, D);
    if ((rc = launch_check(h, "ipm iteration"))) return rc;
  }
  // results: x back into the caller's array, multipliers, per-instance verdicts
  hipLaunchKernelGGL(ipm_pack_x_kernel, gx, dim3(256), 0, st, D);
  IPM_TRY(h, hipMemcpyAsync(d_x, D.xe, size_t(B) * p.n * sizeof(double), hipMemcpyDeviceToDevice, st));
  if (d_lambda) IPM_TRY(h, hipMemcpyAsync(d_lambda, D.lam, size_t(B) * p.m * sizeof(double), hipMemcpyDeviceToDevice, st));
  IPM_TRY(h, hipMemcpyAsync(h->h_inst.data(), D.inst, size_t(B) * sizeof(IpmInst), hipMemcpyDeviceToHost, st));
  IPM_TRY(h, hipStreamSynchronize(st));
  for (unsigned bi = 0; bi < B; ++bi) {
    const IpmInst& S = h->h_inst[bi];
    if (obj) obj[bi] = S.f;
    if (status) status[bi] = S.status == 1 ? 0 : (S.status == 6 ? 1 : S.status);
    if (iterations) iterations[bi] = S.iter;
    if (kkt_error) kkt_error[bi] = S.err0;
  }
  return RPM_OK;
}

int rpm_ipm_solve(rpm_ipm* h, double* x, double* lambda, double* obj, int* status, int* iterations, double* kkt_error) {
  if (!h || !x) return RPM_E_INVALID;
  IpmDev& D = h->D;
  const IpmPlan& p = h->plan;
  double *d_x = nullptr, *d_l = nullptr;
  IPM_TRY(h, hipMalloc(reinterpret_cast<void**>(&d_x), size_t(D.B) * p.n * sizeof(double)));
  if (hipMalloc(reinterpret_cast<void**>(&d_l), size_t(D.B) * std::max(p.m, 1) * sizeof(double)) != hipSuccess) {
    (void)hipFree(d_x);
    h->err = "hipMalloc";
    return RPM_E_DEVICE;
  }
  int rc = RPM_OK;
  if (hipMemcpy(d_x, x, size_t(D.B) * p.n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) rc = RPM_E_DEVICE;
  if (!rc) rc = rpm_ipm_solve_dev(h, d_x, d_l, obj, status, iterations, kkt_error);
  if (!rc && hipMemcpy(x, d_x, size_t(D.B) * p.n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = RPM_E_DEVICE;
  if (!rc && lambda && hipMemcpy(lambda, d_l, size_t(D.B) * p.m * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = RPM_E_DEVICE;
  (void)hipFree(d_x);
  (void)hipFree(d_l);
  if (rc == RPM_E_DEVICE && h->err.empty()) h->err = "hip copy failed";
  return rc;
}

}  // extern "C"

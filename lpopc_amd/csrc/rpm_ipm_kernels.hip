// rpm_ipm_kernels.hip — row f-2: batched primal-dual interior-point iterations with every iterate, multiplier, KKT
// matrix and factor resident in HBM (see rpm_ipm.hpp for what is restated and what is not).  One workgroup per
// instance for the vector work and for the band + border LDL^T; the NLP callbacks are the engine's own batched launches.
//
// Per iteration (Waechter & Biegler 2006, the equation numbers below are that paper's):
//   grad f + A^T lambda; residuals, optimality error E_0 / E_mu (5), barrier update (7) or the          ipm_jt_lambda_kernel,
//     adaptive rule, tau (8); in restoration mode the same for the restoration problem + the test to leave it   ipm_residual_kernel
//   W = eval_h(x, 1, lambda); K = [[W + Sigma + dw I, A^T], [A, -dc I]] (13)     ipm_assemble_kernel
//     (restoration: [[zeta D_R^2 + Sigma, A^T], [A, -(Sigma_p^-1 + Sigma_n^-1)]]; least-squares multipliers: [[I, A^T], [A, -dc I]])
//   LDL^T without pivoting (one band, or nested dissection over the mesh intervals and, on long meshes, over groups of their
//     separators) + inertia check / correction (Algorithm IC)                    kkt_factor_kernel, kkt_gather_add_kernel, ipm_inertia_kernel
//   direction, dz (12), fraction to the boundary (15), alpha_min (23)            kkt_solve_kernel, kkt_vec_kernel, ipm_direction_kernel
//   filter line search (18)-(20), (22), second-order correction (A-5.5 .. 5.9)   ipm_trial_kernel, ipm_accept_kernel, ipm_soc_rhs_kernel, ipm_soc_direction_kernel
//   step, multiplier reset (16), filter update, entry into the restoration phase ipm_update_kernel
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>

#include "rpm_device_internal.hpp"
#include "rpm_ipm_device.hpp"

namespace rpm {


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
// value of `v` in lane `lane` of the wave (lane uniform; a constant after unrolling): two v_readlane_b32 instead of the
// ds_bpermute pair of __shfl
__device__ inline double readlane_d(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
// value of `v` in the lane whose number is byte_addr / 4 (any lane per lane): a ds_bpermute pair
__device__ inline double bperm_d(int byte_addr, double v) {
  const int lo = __builtin_amdgcn_ds_bpermute(byte_addr, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(byte_addr, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
// value of `v` in lane K of the reader's own row of 16 lanes (DPP row_newbcast)
template <int K>
__device__ inline double row_bcast_d(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x150 + K, 0xf, 0xf, false), hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x150 + K, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// One step of the 16 x 16 LDL^T + inverse of kkt_factor_dense_kernel's diagonal wave with the block's COLUMNS dealt to the four
// rows of 16 lanes: lane (q, r) = 16 q + r keeps, of matrix row r, the columns 4 g + q (R[g]: the block, V[g]: L11^-1 so far).
// Step K: the pivot column sits in group K & 3 — every lane fetches its own row's entry of it (the multiplier's numerator) and
// the entries of the rows that are its columns (ds_bpermute, 10 a step), row K of the inverse comes by row_newbcast, and a lane
// is left with 4 or 5 of the 16 products of a step.  Every entry sees the same operations in the same order as with a whole
// row per lane (a wave-uniform v_readlane per operand, 34 a step, 16 dependent products).
template <int K>
struct DiagStep {
  static __device__ __forceinline__ void run(double (&R)[4], double (&V)[4], int q, int r, int a_own, const int (&a_col)[4]) {
    constexpr int QK = K & 3, GK = K >> 2;
    const double dk = readlane_d(R[GK], QK * 16 + K);
    const double ar = bperm_d(a_own + QK * 64, R[GK]);
    double aj[4] = {0.0, 0.0, 0.0, 0.0}, mk[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (4 * g + 3 > K) aj[g] = bperm_d(a_col[g] + QK * 64, R[GK]);     // a(4 g + q, K) before scaling
      if (4 * g <= K) mk[g] = row_bcast_d<K>(V[g]);                       // row K of the inverse so far
    }
    const double lik = r > K ? ar / dk : 0.0;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (4 * g > K) R[g] = __builtin_fma(-lik, aj[g], R[g]);             // columns right of the pivot
      else if (4 * g + 3 > K) { const double u = __builtin_fma(-lik, aj[g], R[g]); R[g] = q > K - 4 * g ? u : R[g]; }
      if (4 * g + 3 <= K) V[g] = __builtin_fma(-lik, mk[g], V[g]);        // columns up to the pivot
      else if (4 * g <= K) { const double u = __builtin_fma(-lik, mk[g], V[g]); V[g] = q <= K - 4 * g ? u : V[g]; }
    }
    if (q == QK && r > K) R[GK] = lik;
    if constexpr (K + 1 < IPM_W) DiagStep<K + 1>::run(R, V, q, r, a_own, a_col);
  }
};
__device__ inline bool has_lo(double l, double u) { return l > -IPM_INF && l != u; }
__device__ inline bool has_up(double l, double u) { return u < IPM_INF && l != u; }

// Several workgroups per instance (gridDim.x = G > 1; blockIdx.y = instance) when a few large instances run: every workgroup
// takes a slice and leaves its N partial results (kind 0 sum, 1 max, 2 min) in D.part; the LAST one to arrive — a ticket in
// D.tick — combines them in workgroup order, so the totals do not depend on who finishes last, and carries on alone with the
// instance's verdicts (true is returned on that workgroup only; with G = 1 always).  Until then nobody has changed the
// instance record, so every workgroup has read the same flags.
template <int N>
__device__ inline bool vec_combine(const IpmDev& D, int bi, double (&vals)[N], const int (&kind)[N]) {
  static_assert(N <= IPM_VEC_PART, "IPM_VEC_PART");
  const int G = gridDim.x;
  if (G == 1) return true;
  __shared__ int last_arrival;
  double* P = D.part + size_t(bi) * IPM_VEC_BLOCKS * IPM_VEC_PART;
  if (threadIdx.x == 0) {
    double* mine = P + size_t(blockIdx.x) * IPM_VEC_PART;
#pragma unroll
    for (int k = 0; k < N; ++k) mine[k] = vals[k];
    __threadfence();
    last_arrival = atomicAdd(&D.tick[bi], 1) == G - 1;
  }
  __syncthreads();
  if (!last_arrival) return false;
  __threadfence();
  // the partial results into LDS side by side (as G x N dependent loads of every thread they took 20 us of a 30 us kernel on the
  // metric problem), then added up in workgroup order as before
  __shared__ double staged[IPM_VEC_BLOCKS * N];
  for (int idx = threadIdx.x; idx < G * N; idx += blockDim.x) staged[idx] = P[size_t(idx / N) * IPM_VEC_PART + idx % N];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < N; ++k) vals[k] = kind[k] == 0 ? 0.0 : (kind[k] == 1 ? -1e300 : 1e300);
  for (int b = 0; b < G; ++b) {
    const double* q = staged + b * N;
#pragma unroll
    for (int k = 0; k < N; ++k) vals[k] = kind[k] == 0 ? vals[k] + q[k] : (kind[k] == 1 ? fmax(vals[k], q[k]) : fmin(vals[k], q[k]));
  }
  if (threadIdx.x == 0) D.tick[bi] = 0;
  return true;
}

// ------------------------------------------------------------------------------------------------ start
// x pushed into the interior of its bounds (Ipopt 3.12 bound_push / bound_frac, paper section 3.6), z = 1, lambda = 0.  The
// bounds themselves first move out by bound_relax * max(1, |bound|) (Ipopt's bound_relax_factor): vl0 / vu0 keep the caller's.
__device__ inline double push_inside(double x, double l, double u, bool lo, bool up, const IpmOpts& o) {
  if (lo) {
    const double p = up ? fmin(o.bound_push * fmax(1.0, fabs(l)), o.bound_frac * (u - l)) : o.bound_push * fmax(1.0, fabs(l));
    x = fmax(x, l + p);
  }
  if (up) {
    const double p = lo ? fmin(o.bound_push * fmax(1.0, fabs(u)), o.bound_frac * (u - l)) : o.bound_push * fmax(1.0, fabs(u));
    x = fmin(x, u - p);
  }
  return x;
}
__global__ __launch_bounds__(256) void ipm_init_kernel(IpmDev D, const double* x0) {
  const int bi = blockIdx.x;
  const size_t o = size_t(bi) * D.nv;
  for (int i = threadIdx.x; i < D.n; i += blockDim.x) {
    double x = x0[size_t(bi) * D.n + i];
    double l = D.vl0[o + i], u = D.vu0[o + i];
    const bool lo = l > -IPM_INF && l != u, up = u < IPM_INF && l != u;
    if (l == u) x = l;
    else {
      if (lo) l -= D.o.bound_relax * fmax(1.0, fabs(l));
      if (up) u += D.o.bound_relax * fmax(1.0, fabs(u));
      x = push_inside(x, l, u, lo, up, D.o);
    }
    D.v[o + i] = x;
    D.vl[o + i] = l;
    D.vu[o + i] = u;
    D.zL[o + i] = lo ? 1.0 : 0.0;
    D.zU[o + i] = up ? 1.0 : 0.0;
  }
  for (int r = threadIdx.x; r < D.m; r += blockDim.x) D.lam[size_t(bi) * D.m + r] = 0.0;
  if (threadIdx.x == 0) {
    IpmInst& S = D.inst[bi];
    S = IpmInst{};
    S.mu = D.o.mu_init;
    if (D.o.init_ls_mult && D.m > 0) { S.mode = 3; S.skip_update = -2; }   // first pass: least-squares multipliers at the starting point
  }
}
// slacks start at g(x0), pushed inside the (relaxed) [g_l, g_u] the same way
__global__ __launch_bounds__(256) void ipm_init_slack_kernel(IpmDev D) {
  const int bi = blockIdx.x;
  for (int s = threadIdx.x; s < D.ns; s += blockDim.x) {
    const int r = D.slack_row[s];
    double l = D.gl[r], u = D.gu[r];
    const bool lo = l > -IPM_INF, up = u < IPM_INF;
    if (D.scal_on) {           // the rows are scaled (nlp_scaling): so are their bounds
      const double sr = D.sc[size_t(bi) * D.m + r];
      if (lo) l *= sr;
      if (up) u *= sr;
    }
    if (lo) l -= D.o.bound_relax * fmax(1.0, fabs(l));
    if (up) u += D.o.bound_relax * fmax(1.0, fabs(u));
    const size_t o = size_t(bi) * D.nv + D.n + s;
    D.v[o] = push_inside(D.g[size_t(bi) * D.sg + r], l, u, lo, up, D.o);
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
// multiplier reset (16) of a bound multiplier z for the slack s
__device__ inline double reset16(double z, double s, double mu, double ks) { return fmax(fmin(z, ks * mu / s), mu / (ks * s)); }

// grad f + A^T lambda by column (the restoration problem has no grad f).  A thread per unknown for the short columns; the long
// ones — t0, tf, the final states: every defect row of a phase, 7 168 entries on the metric problem, which one thread walked
// in 1.4 ms — take a workgroup each (blockIdx.x >= the thread-per-unknown blocks), summed in a fixed order.
constexpr int IPM_LONG_COLUMN = 256;
__global__ __launch_bounds__(256) void ipm_jt_lambda_kernel(IpmDev D, int n_thread_blocks) {
  __shared__ double sh[16];
  const int bi = blockIdx.y;
  const IpmInst& S = D.inst[bi];
  if (S.status != 0) return;
  const double *lam = D.lam + size_t(bi) * D.m, *jac = D.jac + size_t(bi) * D.sv;
  const bool resto = S.mode == 2;
  if (int(blockIdx.x) >= n_thread_blocks) {
    const int i = D.long_cols[blockIdx.x - n_thread_blocks];
    double acc = 0.0;
    for (int q = D.jt_ptr[i] + threadIdx.x; q < D.jt_ptr[i + 1]; q += blockDim.x) acc += jac[D.jt_ent[q]] * lam[D.jt_row[q]];
    acc = block_red(acc, 0, sh);
    if (threadIdx.x == 0) D.glag[size_t(bi) * D.nv + i] = (resto ? 0.0 : D.grad[size_t(bi) * D.n + i]) + acc;
    return;
  }
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= D.nv) return;
  double acc;
  if (i < D.n) {
    if (D.jt_ptr[i + 1] - D.jt_ptr[i] > IPM_LONG_COLUMN) return;
    acc = resto ? 0.0 : D.grad[size_t(bi) * D.n + i];
    for (int q = D.jt_ptr[i]; q < D.jt_ptr[i + 1]; ++q) acc += jac[D.jt_ent[q]] * lam[D.jt_row[q]];
  } else {
    acc = -lam[D.slack_row[i - D.n]];
  }
  D.glag[size_t(bi) * D.nv + i] = acc;
}
__global__ __launch_bounds__(1024) void ipm_residual_kernel(IpmDev D) {
  __shared__ double sh[16];
  __shared__ int verdict;       // restoration: 0 stay, 1 leave it (least-squares multipliers next), 2 stop
  // (several workgroups per instance when a few large instances run: slices i0, i0 + stride, ...; vec_combine)
  const int bi = blockIdx.y, t = threadIdx.x, i0 = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  IpmInst& S = D.inst[bi];
  if (S.status != 0) return;
  const int mode_in = S.mode;
  if (mode_in == 3) {           // recalc_y: a pass that only recomputes the multipliers at this point (set by ipm_update_kernel)
    if (t == 0 && blockIdx.x == 0) { S.refactor = 1; S.delta_w = 0.0; atomicAdd(&D.cnt[0], 1); }
    return;
  }
  const double *v = D.v + size_t(bi) * D.nv, *vl = D.vl + size_t(bi) * D.nv, *vu = D.vu + size_t(bi) * D.nv;
  const double *zL = D.zL + size_t(bi) * D.nv, *zU = D.zU + size_t(bi) * D.nv, *lam = D.lam + size_t(bi) * D.m;
  const double *g = D.g + size_t(bi) * D.sg, *glag = D.glag + size_t(bi) * D.nv;
  double dinf = 0, cinf = 0, th1 = 0, cmax = 0, cmin = 1e300, sl = 0, sz = 0, ln = 0, bad = 0, nzb = 0;
  double csq = 0, dsq = 0, psum = 0, psq = 0, nfree = 0;      // 2-norms for the adaptive barrier update's KKT error
  double cinf_u = 0;            // nlp_scaling: the constraint violation of the unscaled problem (Ipopt's constr_viol_tol applies to it)
  // pass 1: constraint values and what does not depend on the multipliers
  #pragma unroll 4
  for (int r = i0; r < D.m; r += stride) {
    const int s = D.row_slack[r];
    const double glr = D.scal_on ? D.sc[size_t(bi) * D.m + r] * D.gl[r] : D.gl[r];
    const double cr = s < 0 ? g[r] - glr : g[r] - v[D.n + s];
    D.c[size_t(bi) * D.m + r] = cr;
    if (!(fabs(cr) < 1e300)) bad = 1;
    cinf = fmax(cinf, fabs(cr));
    cinf_u = fmax(cinf_u, D.scal_on ? fabs(cr / D.sc[size_t(bi) * D.m + r]) : fabs(cr));
    th1 += fabs(cr);
    csq += cr * cr;
  }
  cinf = block_red(cinf, 1, sh); th1 = block_red(th1, 0, sh); cinf_u = block_red(cinf_u, 1, sh);
  if (mode_in == 2) {
    // ---- restoration phase (paper section 3.3): min rho sum(p + n) + zeta/2 |D_R (v - v_R)|^2  s.t.  c(v) - p + n = 0, p, n >= 0, bounds
    const IpmOpts& o = D.o;
    const double rho = o.resto_rho, zeta = S.zeta;
    const double *pp = D.pp + size_t(bi) * D.m, *nn = D.nn + size_t(bi) * D.m, *zp = D.zp + size_t(bi) * D.m, *zn = D.zn + size_t(bi) * D.m;
    const double *vR = D.vR + size_t(bi) * D.nv, *dr2 = D.dr2 + size_t(bi) * D.nv;
    double thr = 0, rinf = 0, spn = 0, lnpn = 0, qd = 0;
    #pragma unroll 4
    for (int r = i0; r < D.m; r += stride) {
      const double rc = D.c[size_t(bi) * D.m + r] - pp[r] + nn[r];
      thr += fabs(rc); rinf = fmax(rinf, fabs(rc));
      spn += pp[r] + nn[r];
      lnpn += log(pp[r]) + log(nn[r]);
      dinf = fmax(dinf, fmax(fabs(rho - lam[r] - zp[r]), fabs(rho + lam[r] - zn[r])));
      const double p1 = zp[r] * pp[r], p2 = zn[r] * nn[r];
      cmax = fmax(cmax, fmax(p1, p2)); cmin = fmin(cmin, fmin(p1, p2));
    }
    #pragma unroll 4
    for (int i = i0; i < D.nv; i += stride) {
      const double acc = glag[i];             // A^T lambda only (ipm_jt_lambda_kernel): the proximity term is added where zeta is known
      const double l = vl[i], u = vu[i];
      if (l == u) continue;
      const double dd = v[i] - vR[i];
      qd += dr2[i] * dd * dd;
      dinf = fmax(dinf, fabs(zeta * dr2[i] * dd + acc - zL[i] + zU[i]));
      if (!(fabs(acc) < 1e300)) bad = 1;
      if (l > -IPM_INF) { const double d = v[i] - l, pr = zL[i] * d; cmax = fmax(cmax, pr); cmin = fmin(cmin, pr); ln += log(d); }
      if (u < IPM_INF) { const double d = u - v[i], pr = zU[i] * d; cmax = fmax(cmax, pr); cmin = fmin(cmin, pr); ln += log(d); }
    }
    thr = block_red(thr, 0, sh); rinf = block_red(rinf, 1, sh); spn = block_red(spn, 0, sh); lnpn = block_red(lnpn, 0, sh);
    qd = block_red(qd, 0, sh); dinf = block_red(dinf, 1, sh); cmax = block_red(cmax, 1, sh); cmin = block_red(cmin, 2, sh);
    ln = block_red(ln, 0, sh); bad = block_red(bad, 1, sh);
    {
      double vals[12] = {thr, rinf, spn, lnpn, qd, dinf, cmax, cmin, ln, bad, cinf, th1};
      const int kind[12] = {0, 1, 0, 0, 0, 1, 1, 2, 0, 1, 1, 0};
      if (!vec_combine(D, bi, vals, kind)) return;
      thr = vals[0]; rinf = vals[1]; spn = vals[2]; lnpn = vals[3]; qd = vals[4]; dinf = vals[5]; cmax = vals[6]; cmin = vals[7];
      ln = vals[8]; bad = vals[9]; cinf = vals[10]; th1 = vals[11];
    }
    if (t == 0) {
      const double f = D.obj[bi];
      verdict = 0;
      if (bad != 0 || !(fabs(f) < 1e300) || !(fabs(ln) < 1e300) || !(fabs(lnpn) < 1e300)) { S.status = 5; verdict = 2; }
      else {
        const double phi_o = f - S.mu * ln;      // the ORIGINAL barrier objective: what the original filter is asked about
        bool back = S.resto_it > 0 && th1 <= o.kappa_resto * S.th0 && th1 <= S.theta_max;
        const double* F = D.filt + size_t(bi) * 2 * IPM_FMAX;
        for (int k = 0; back && k < S.nfilt; ++k)
          if (th1 >= F[2 * k] && phi_o >= F[2 * k + 1]) back = false;
        if (back) verdict = 1;
        else if (S.resto_it >= o.resto_max) { S.status = 3; verdict = 2; }
        else if (S.iter >= o.max_iter) { S.status = 2; verdict = 2; }
        else {
          if (S.resto_it == 0) { S.thr_max = 1e4 * fmax(1.0, thr); S.thr_min = 1e-4 * fmax(1.0, thr); }
          const double mu_min = o.tol / 10.0;
          double mu_r = S.mu_r;
          bool stuck = false;
          for (int guard = 0; guard < 64; ++guard) {
            const double emu = fmax(fmax(dinf, rinf), fmax(fabs(cmax - mu_r), fabs(cmin - mu_r)));
            if (!(emu <= o.kappa_eps * mu_r)) break;
            if (mu_r <= mu_min) { stuck = true; break; }     // a minimiser of the infeasibility that the filter does not take
            mu_r = fmax(mu_min, fmin(o.kappa_mu * mu_r, pow(mu_r, o.theta_mu)));
            S.nrfilt = 0;
          }
          if (stuck) { S.status = 3; verdict = 2; }
          else {
            S.mu_r = mu_r; S.zeta = sqrt(mu_r); S.tau = fmax(o.tau_min, 1.0 - mu_r);
            S.f = f; S.theta = th1; S.lnsum = ln; S.cinf = cinf; S.th_r = thr;
            S.phi_r = rho * spn + 0.5 * S.zeta * qd - mu_r * (ln + lnpn);
            S.refactor = 1;
            S.delta_w = 0.0;
            atomicAdd(&D.cnt[0], 1);
          }
        }
      }
    }
    __syncthreads();
    if (verdict != 1) return;
    // leaving the restoration: bound multipliers clipped against the ORIGINAL mu, then one pass that only computes
    // least-squares multipliers (mode 3; paper section 3.6) before the regular iteration resumes at this point
    #pragma unroll 4
    for (int i = t; i < D.nv; i += blockDim.x) {
      const double l = vl[i], u = vu[i];
      if (l == u) continue;
      const size_t o2 = size_t(bi) * D.nv + i;
      if (l > -IPM_INF) D.zL[o2] = reset16(fmin(D.zL[o2], 1e3), v[i] - l, S.mu, D.o.kappa_sigma);
      if (u < IPM_INF) D.zU[o2] = reset16(fmin(D.zU[o2], 1e3), u - v[i], S.mu, D.o.kappa_sigma);
    }
    if (t == 0) { S.mode = 3; S.refactor = 1; S.delta_w = 0.0; atomicAdd(&D.cnt[0], 1); }
    return;
  }
  // pass 2: gradient of the Lagrangian, complementarity products
  #pragma unroll 4
  for (int i = i0; i < D.nv; i += stride) {
    const double acc = glag[i];               // grad f + A^T lambda (ipm_jt_lambda_kernel)
    const double l = vl[i], u = vu[i], vi = v[i], zli = zL[i], zui = zU[i];    // loads ahead of the branch
    if (l != u) {
      const double dres = acc - zli + zui;
      dinf = fmax(dinf, fabs(dres));
      dsq += dres * dres; nfree += 1;
      if (!(fabs(acc) < 1e300)) bad = 1;
      if (l > -IPM_INF) {
        const double d = vi - l, pr = zli * d;
        cmax = fmax(cmax, pr); cmin = fmin(cmin, pr); sz += zli; ln += log(d); nzb += 1; psum += pr; psq += pr * pr;
      }
      if (u < IPM_INF) {
        const double d = u - vi, pr = zui * d;
        cmax = fmax(cmax, pr); cmin = fmin(cmin, pr); sz += zui; ln += log(d); nzb += 1; psum += pr; psq += pr * pr;
      }
    }
  }
  csq = block_red(csq, 0, sh); dsq = block_red(dsq, 0, sh); psum = block_red(psum, 0, sh); psq = block_red(psq, 0, sh);
  nfree = block_red(nfree, 0, sh);
  #pragma unroll 4
  for (int r = i0; r < D.m; r += stride) sl += fabs(lam[r]);
  dinf = block_red(dinf, 1, sh);
  cmax = block_red(cmax, 1, sh); cmin = block_red(cmin, 2, sh); sl = block_red(sl, 0, sh); sz = block_red(sz, 0, sh);
  ln = block_red(ln, 0, sh); bad = block_red(bad, 1, sh); nzb = block_red(nzb, 0, sh);
  {
    double vals[16] = {dinf, cmax, cmin, sl, sz, ln, bad, nzb, cinf, th1, csq, dsq, psum, psq, nfree, cinf_u};
    const int kind[16] = {1, 1, 2, 0, 0, 0, 1, 0, 1, 0, 0, 0, 0, 0, 0, 1};
    if (!vec_combine(D, bi, vals, kind)) return;
    dinf = vals[0]; cmax = vals[1]; cmin = vals[2]; sl = vals[3]; sz = vals[4]; ln = vals[5]; bad = vals[6]; nzb = vals[7];
    cinf = vals[8]; th1 = vals[9]; csq = vals[10]; dsq = vals[11]; psum = vals[12]; psq = vals[13]; nfree = vals[14]; cinf_u = vals[15];
  }
  if (t != 0) return;
  const IpmOpts& o = D.o;
  S.f = D.obj[bi]; S.theta = th1; S.lnsum = ln; S.dinf = dinf; S.cinf = cinf; S.comp_max = cmax; S.comp_min = cmin;
  S.sum_lam = sl; S.sum_z = sz; S.nzb = int(nzb);
  if (!(fabs(S.f) < 1e300) || !(fabs(ln) < 1e300)) bad = 1;
  const double sd = fmax(o.s_max, (sl + sz) / fmax(1.0, double(D.m) + nzb)) / o.s_max;   // (6)
  const double sc = nzb > 0 ? fmax(o.s_max, sz / nzb) / o.s_max : 1.0;
  S.err0 = fmax(fmax(dinf / sd, cinf), nzb > 0 ? cmax / sc : 0.0);
  if (bad != 0) { S.status = 5; return; }
  // Ipopt's secondary thresholds apply to the unscaled problem: gradient of the Lagrangian and complementarity / sf, rows / sc
  const double sfu = D.scal_on ? D.sf[bi] : 1.0;
  const double cm = (nzb > 0 ? cmax : 0.0) / sfu, dinf_u = dinf / sfu;
  if (S.err0 <= o.tol && dinf_u <= o.dual_inf_tol && cinf_u <= o.constr_viol_tol && cm <= o.compl_inf_tol) { S.status = 1; return; }
  S.n_acc = (S.err0 <= o.acceptable_tol && dinf_u <= o.acc_dual_inf_tol && cinf_u <= o.acc_constr_viol_tol && cm <= o.acc_compl_inf_tol) ? S.n_acc + 1 : 0;
  if (o.acceptable_iter > 0 && S.n_acc >= o.acceptable_iter) { S.status = 6; return; }
  if (S.iter >= o.max_iter) { S.status = 2; return; }
  if (S.iter == 0) {
    S.theta_max = 1e4 * fmax(1.0, th1);
    S.theta_min = 1e-4 * fmax(1.0, th1);
  }
  const double mu_min = o.tol / 10.0;
  double mu = S.mu;
  bool from_oracle = false;
  if (o.mu_adaptive && nzb > 0) {      // Ipopt's adaptive update: LOQO oracle, kkt-error globalisation (oracle/ipm_oracle.py)
    const double avg = psum / nzb;
    if (S.mu_max == 0.0) S.mu_max = o.mu_max_fact * avg;
    const double kkt_err = dsq / fmax(1.0, nfree) + (D.m ? csq / D.m : 0.0) + psq / nzb;
    bool progress = S.nrefs < 4;
    for (int k = 0; !progress && k < S.nrefs; ++k) progress = kkt_err <= o.mu_red_fact * S.refs[k];
    double mu_new = -1.0;
    if (progress) {
      S.fixed_mode = 0;
      if (S.nrefs < 4) S.refs[S.nrefs++] = kkt_err;
      else { S.refs[0] = S.refs[1]; S.refs[1] = S.refs[2]; S.refs[2] = S.refs[3]; S.refs[3] = kkt_err; }
      const double xi = cmin / avg, fac = fmin(0.05 * (1.0 - xi) / xi, 2.0);
      mu_new = fmax(mu_min, fmin(0.1 * fac * fac * fac * avg, S.mu_max));
    } else if (!S.fixed_mode) {        // no progress in the free mode: the monotone rule takes over from here
      S.fixed_mode = 1;
      mu_new = fmax(mu_min, fmin(o.mu_init_factor * avg, S.mu_max));
    }
    if (mu_new >= 0.0) {
      if (mu_new != mu) S.nfilt = 0;
      mu = mu_new;
      from_oracle = true;
    }
  }
  for (int guard = 0; !from_oracle && guard < 64; ++guard) {
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
  if (o.ic_hot && S.ic_hot && o.kw_dec * S.delta_w_last >= o.ic_hot_min) S.delta_w = o.kw_dec * S.delta_w_last;
  atomicAdd(&D.cnt[0], 1);
}

// ------------------------------------------------------------------------------------------------ KKT matrix + rhs
__global__ void ipm_zero_kernel(IpmDev D) {
  const int bi = blockIdx.y;
  const IpmInst& S = D.inst[bi];
  if (S.status != 0 || !S.refactor) return;
  double2* K = reinterpret_cast<double2*>(D.K + size_t(bi) * D.kstride);
  const long long n2 = D.kstride / 2;
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
  // mode 0: (13).  mode 2 (restoration, p and n eliminated): [[zeta D_R^2 + Sigma_v, A^T], [A, -(Sigma_p^-1 + Sigma_n^-1)]], no
  // Hessian (Gauss-Newton model).  mode 3 (least-squares multipliers): [[I, A^T], [A, -delta_c]].
  const int mode = S.mode;
  if (mode == 0 && !D.lb_on)   // duplicates of a slot (I-part / E-part of the reference's COO) are summed in COO order: bit-reproducible.  The
    for (int i = t0; i < D.n_hg; i += stride) {   // slot may also take the diagonal term below: two atomic adds onto zero commute
      double acc = 0.0;
      for (int j = D.hg_ptr[i]; j < D.hg_ptr[i + 1]; ++j) acc += D.hess[size_t(bi) * D.nnz_h + D.hg_src[j]];
      unsafeAtomicAdd(&K[D.hg_dst[i]], acc);
    }
  for (int k = t0; k < D.nnz_jac; k += stride)
    if (D.jac_dst[k] >= 0) K[D.jac_dst[k]] = D.jac[size_t(bi) * D.sv + k];
  for (int s = t0; s < D.ns; s += stride) K[D.slk_dst[s]] = -1.0;
  double* rhs = D.rhs + size_t(bi) * D.Nt;
  const double mu = mode == 2 ? S.mu_r : S.mu;
  for (int i = t0; i < D.nv; i += stride) {
    const double l = vl[i], u = vu[i];
    double diag = 1.0, r = 0.0;
    if (l != u) {
      if (mode == 3) {
        diag = 1.0 + S.delta_w;
        r = (i < D.n ? D.grad[size_t(bi) * D.n + i] : 0.0) - zL[i] + zU[i];
      } else {
        diag = S.delta_w;
        if (mode == 0 && D.lb_on && i < D.n) diag += D.lb_small[size_t(bi) * IPM_LB_SMALL];   // sigma I of the limited-memory Hessian
        r = D.glag[size_t(bi) * D.nv + i];
        if (mode == 2) {
          const double w2 = S.zeta * D.dr2[size_t(bi) * D.nv + i];
          diag += w2;
          r += w2 * (v[i] - D.vR[size_t(bi) * D.nv + i]);
        }
        const double cap = D.o.sigma_cap > 0 ? D.o.sigma_cap : 1e300;
        if (l > -IPM_INF) { const double d = v[i] - l; diag += fmin(zL[i] / d, cap); r -= mu / d; }
        if (u < IPM_INF) { const double d = u - v[i]; diag += fmin(zU[i] / d, cap); r += mu / d; }
      }
    }
    unsafeAtomicAdd(&K[D.diag_dst[i]], diag);
    rhs[D.pos[i]] = -r;
  }
  for (int r = t0; r < D.m; r += stride) {
    double k22 = -D.o.delta_c, rr = 0.0;
    if (mode == 0) rr = -D.c[size_t(bi) * D.m + r];
    else if (mode == 2) {
      const size_t q = size_t(bi) * D.m + r;
      const double pp = D.pp[q], nn = D.nn[q], sp = D.zp[q] / pp, sn = D.zn[q] / nn, lam = D.lam[q];
      const double rp = D.o.resto_rho - lam - mu / pp, rn = D.o.resto_rho + lam - mu / nn;
      k22 = -(1.0 / sp + 1.0 / sn);
      rr = -(D.c[q] - pp + nn) - rp / sp + rn / sn;
    }
    K[D.diag_dst[D.nv + r]] = k22;
    rhs[D.pos[D.nv + r]] = rr;
  }
}

// The value of a structural slot of instance bi's KKT matrix in its present mode (ki / hg coded as IpmDev::as_ki / as_hg) and, for
// a diagonal slot, the right-hand side entry that goes with it (written on the way).  Same values as ipm_assemble_kernel's, bit
// for bit: a slot that took two atomic adds onto zero there (Hessian sum, diagonal term) gets their sum.
struct SlotValue {
  const IpmDev& D;
  int bi, mode;
  double mu, delta_w, zeta;
  const double *v, *vl, *vu, *zL, *zU;
  double* rhs;
  __device__ SlotValue(const IpmDev& D_, int bi_) : D(D_), bi(bi_) {
    const IpmInst& S = D.inst[bi];
    mode = S.mode;
    mu = mode == 2 ? S.mu_r : S.mu;
    delta_w = S.delta_w;
    zeta = S.zeta;
    v = D.v + size_t(bi) * D.nv; vl = D.vl + size_t(bi) * D.nv; vu = D.vu + size_t(bi) * D.nv;
    zL = D.zL + size_t(bi) * D.nv; zU = D.zU + size_t(bi) * D.nv;
    rhs = D.rhs + size_t(bi) * D.Nt;
  }
  // entry e of the tables ki_tab / hg_tab (the Hessian slot that shares a diagonal is looked up only for a diagonal)
  __device__ double operator()(const int* ki_tab, const int* hg_tab, int e) const {
    const int ki = ki_tab[e], kind = ki >> 28, idx = ki & 0x0fffffff;
    if (kind == 1) return D.jac[size_t(bi) * D.sv + idx];
    if (kind == 2) return -1.0;
    if (kind == 4) {   // diagonal of constraint row idx
      double k22 = -D.o.delta_c, rr = 0.0;
      if (mode == 0) rr = -D.c[size_t(bi) * D.m + idx];
      else if (mode == 2) {
        const size_t q = size_t(bi) * D.m + idx;
        const double pp = D.pp[q], nn = D.nn[q], sp = D.zp[q] / pp, sn = D.zn[q] / nn, lam = D.lam[q];
        const double rp = D.o.resto_rho - lam - mu / pp, rn = D.o.resto_rho + lam - mu / nn;
        k22 = -(1.0 / sp + 1.0 / sn);
        rr = -(D.c[q] - pp + nn) - rp / sp + rn / sn;
      }
      rhs[D.pos[D.nv + idx]] = rr;
      return k22;
    }
    // 0: a Hessian slot; 3: the diagonal of variable idx (with the Hessian slot that shares it)
    const int hgi = kind == 0 ? idx : hg_tab[e];
    double acc = 0.0;
    if (mode == 0 && hgi >= 0 && !D.lb_on)
      for (int j = D.hg_ptr[hgi]; j < D.hg_ptr[hgi + 1]; ++j) acc += D.hess[size_t(bi) * D.nnz_h + D.hg_src[j]];
    if (kind == 0) return 0.0 + acc;   // as the add onto the zeroed slot gave it (a sum of -0.0 becomes +0.0)
    const int i = idx;
    const double l = vl[i], u = vu[i];
    double diag = 1.0, r = 0.0;
    if (l != u) {
      if (mode == 3) {
        diag = 1.0 + delta_w;
        r = (i < D.n ? D.grad[size_t(bi) * D.n + i] : 0.0) - zL[i] + zU[i];
      } else {
        diag = delta_w;
        if (mode == 0 && D.lb_on && i < D.n) diag += D.lb_small[size_t(bi) * IPM_LB_SMALL];   // sigma I of the limited-memory Hessian
        r = D.glag[size_t(bi) * D.nv + i];
        if (mode == 2) {
          const double w2 = zeta * D.dr2[size_t(bi) * D.nv + i];
          diag += w2;
          r += w2 * (v[i] - D.vR[size_t(bi) * D.nv + i]);
        }
        const double cap = D.o.sigma_cap > 0 ? D.o.sigma_cap : 1e300;
        if (l > -IPM_INF) { const double d = v[i] - l; diag += fmin(zL[i] / d, cap); r -= mu / d; }
        if (u < IPM_INF) { const double d = u - v[i]; diag += fmin(zU[i] / d, cap); r += mu / d; }
      }
    }
    rhs[D.pos[i]] = -r;
    // the two-kernel path adds the two terms onto zero, in either order: acc + diag, exactly
    return (mode == 0 && hgi >= 0) ? acc + diag : diag;
  }
};

// ipm_zero_kernel + ipm_assemble_kernel in one pass over the storage, by destination: a workgroup builds a chunk of the
// storage in LDS (zeros, then the chunk's structural slots, as_* tables in ascending order) and writes it out in full lines,
// so every line of the storage leaves the chip once per refactorisation instead of being zeroed, fetched again for a
// scattered read-modify-write and written a second time (1024-instance quadrotor sweep: 0.59 + 0.72 ms per iteration for the
// two kernels).  Chunks inside a level-1 block that kkt_factor_dense_kernel assembles itself (IpmDev::df_on) are left out.
__global__ __launch_bounds__(256) void ipm_fill_kernel(IpmDev D) {
  __shared__ double buf[IPM_FILL_CHUNK];
  const int bi = blockIdx.y;
  const IpmInst& S = D.inst[bi];
  if (S.status != 0 || !S.refactor) return;
  double* K = D.K + size_t(bi) * D.kstride;
  const int tid = threadIdx.x;
  const SlotValue value(D, bi);
  const bool some = D.df_on && D.l1_dense_lds && D.as_live;     // only the chunks not inside a level-1 block (kkt_level1_fused)
  const int n_here = some ? D.as_nlive : D.as_nchunk;
  for (int ci = blockIdx.x; ci < n_here; ci += gridDim.x) {
    const int c = some ? D.as_live[ci] : ci;
    const long long lo = (long long)c * IPM_FILL_CHUNK, hi = min(lo + IPM_FILL_CHUNK, D.kstride);
    const int e0 = D.as_ptr[c], e1 = D.as_ptr[c + 1];
    // this thread's first slot: its loads are under way while the chunk is zeroed
    const int ef = e0 + tid;
    double val_f = 0.0;
    int off_f = -1;
    if (ef < e1) { off_f = int(D.as_dst[ef] - lo); val_f = value(D.as_ki, D.as_hg, ef); }
    double2* b2 = reinterpret_cast<double2*>(buf);
    for (int i = tid; i < IPM_FILL_CHUNK / 2; i += 256) b2[i] = make_double2(0.0, 0.0);
    __syncthreads();
    if (off_f >= 0) buf[off_f] = val_f;
    for (int e = ef + 256; e < e1; e += 256) buf[D.as_dst[e] - lo] = value(D.as_ki, D.as_hg, e);
    __syncthreads();
    const int len = int(hi - lo);
    if ((reinterpret_cast<size_t>(K + lo) & 15) == 0) {
      double2* K2 = reinterpret_cast<double2*>(K + lo);
      typedef double d2v __attribute__((ext_vector_type(2)));   // streaming stores: the storage is next read by the factorisation, from HBM either way
      for (int i = tid; i < (len >> 1); i += 256) __builtin_nontemporal_store(d2v{b2[i].x, b2[i].y}, reinterpret_cast<d2v*>(K2 + i));
      if ((len & 1) && tid == 0) K[hi - 1] = buf[len - 1];
    } else {
      for (int i = tid; i < len; i += 256) K[lo + i] = buf[i];
    }
    __syncthreads();   // buf is free for the next chunk
  }
}

// ------------------------------------------------------------------------------------------------ band + border LDL^T
// Storage of one instance: column j holds rows j .. j+b of the band (slot i-j) and the nb border rows (slot b+1+i-Nb).
// IPM_W columns at a time: the diagonal block is factored in LDS, each panel row is solved by the thread that owns it.
// No pivoting: with dw large enough and dc > 0 the matrix is symmetric quasi-definite, whose LDL^T exists for every
// ordering (Vanderbei 1995); the signs of D give the inertia Algorithm IC asks for.
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
#ifndef IPM_DIAG_SPLIT
#define IPM_DIAG_SPLIT 1   // kkt_factor_dense_kernel's diagonal blocks over all 64 lanes of their wave (0: a whole row per lane, four copies)
#endif
#ifndef IPM_LB
#define IPM_LB 2   // waves per SIMD the 4-tile factorisation is compiled for (3: 168 VGPRs, 52 of them spilled since the pivots moved to v_readlane)
#endif
// MT 16-row tiles per wave, NW waves: <4,4> block columns of up to 256 rows, 3 workgroups per CU; <6,4> 384 rows, 2 per CU (at 3 it
// spills 91 VGPRs); <8,4> 512 rows; <3,8> 384 rows over 8 waves — for a few large instances (the metric problem: 268 workgroups
// on 256 CUs), where a second wave per SIMD halves every wave's share of a block column
template <int MT, int NW = 4>
__global__ __launch_bounds__(64 * NW, NW == 8 ? (MT == 2 ? 4 : 1) : (MT == 4 ? IPM_LB : (MT == 6 ? 2 : 1))) void kkt_factor_kernel(double* Kall, long long kstride, const KktSub* subs, int sub0,
                                                                               int n_here, int n_sub, IpmInst* inst, int* piv, int partial) {
  // workgroup = (instance, sub-problem sub0 + s).  partial 1: nested dissection level 1 — eliminate the band part only and
  // leave the Schur complement of the border x border corner, unfactored, in the corner's storage.  partial 2: the corner only
  // (what kkt_factor_dense_kernel left of a last level whose band part it eliminated; its pivot counts are added to).
  typedef double d4 __attribute__((ext_vector_type(4)));
  constexpr int W = IPM_W;
  const int bi = blockIdx.x / n_here, sidx = sub0 + int(blockIdx.x) % n_here, t = threadIdx.x, nt = blockDim.x;
  const IpmInst& S = inst[bi];
  if (S.status != 0 || !S.refactor) return;
  const KktSub sub = subs[sidx];
  const KktGeom G = sub.g;
  double* K = Kall + size_t(bi) * kstride + sub.koff;
  extern __shared__ double lds[];
  double* T = lds;                              // (b + 24) x W
  double* Dg = T + size_t(G.b + 24) * W;        // W x (W + 1)
  double* Mi = Dg + W * (W + 1);                // W x W: L11^-1, row-major
  double* invd = Mi + W * W;                    // W
  double* BL = invd + W;                        // nb x W: L of the border rows in the current block column
  double* BY = BL + size_t(G.nb) * W;           // nb x W: L D
  double* C = BY + size_t(G.nb) * W;            // nb (nb + 1) / 2: the corner's lower triangle, packed by rows
  __shared__ int cnt[3];
  if (t < 3) cnt[t] = 0;
  int npos = 0, nneg = 0, nbad = 0;
  const int nb = G.nb;
  const int wv = t >> 6, lr = t & 15, lq = (t & 63) >> 4;
#ifdef IPM_TIMING
#ifndef IPM_TIMING_SUB
#define IPM_TIMING_SUB (-1)   // which sub-problem's workgroup reports its phase clocks (-1: the last level's)
#endif
  long long tc[6] = {0, 0, 0, 0, 0, 0}, t_prev = wall_clock64();
#define IPM_TICK(i) do { __syncthreads(); const long long _n = wall_clock64(); tc[i] += _n - t_prev; t_prev = _n; } while (0)
#else
#define IPM_TICK(i)
#endif
  auto ct = [](int r, int c) { return r * (r + 1) / 2 + c; };   // the corner's lower triangle, packed by rows (c <= r)
  for (int idx = t; idx < nb * nb; idx += nt) {
    const int r = idx / nb, c = idx % nb;
    if (r >= c) C[ct(r, c)] = K[G.at(G.Nb + r, G.Nb + c)];
  }
  // corner -= L_border D L_border^T of a block column: the 16 x 16 tiles of its lower triangle dealt to the waves w0, w0 + 1, ..
  // (nw of them), four matrix products each (the scalar form — 16 FMAs per entry fed by LDS reads 128 bytes apart, a 16-way bank
  // conflict — was 73 % of a separator group's factorisation and 23 % of an interval's).  It only needs the column's border rows of
  // L and L D (BL, BY), which stay in LDS until the NEXT panel is solved: so block column J's update runs on the waves that are
  // idle while wave 0 factors the diagonal block of J + 1 (it was a phase of its own, 15 % of an interval block of the metric problem).
  auto corner_update = [&](int w0, int nw) {
    const int nbt = (nb + 15) >> 4, ntl = nbt * (nbt + 1) / 2;
    for (int tl = wv - w0; tl < ntl; tl += nw) {
      int ti = 0;
      while ((ti + 1) * (ti + 2) / 2 <= tl) ++ti;
      const int tj = tl - ti * (ti + 1) / 2, ra = ti * 16 + lr, cb = tj * 16 + lr;
      d4 cacc;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int rr = ti * 16 + lq + 4 * g;
        cacc[g] = (rr < nb && cb <= rr) ? C[ct(rr, cb)] : 0.0;
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const double a = ra < nb ? -BL[ra * W + 4 * g + lq] : 0.0, bq = cb < nb ? BY[cb * W + 4 * g + lq] : 0.0;
        cacc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bq, cacc, 0, 0, 0);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int rr = ti * 16 + lq + 4 * g;
        if (rr < nb && cb <= rr) C[ct(rr, cb)] = cacc[g];
      }
    }
  };
  for (int J0 = partial == 2 ? G.Nb : 0; J0 < G.Nb;) {
    const int J1 = min(J0 + W, G.Nb), w = J1 - J0;
    const int kbase = max(J0 - G.b, 0), nk = J0 - kbase, ngrp = (nk + 3) >> 2;
    const int nrb = max(min(J1 - 1 + G.b, G.Nb - 1) - J1 + 1, 0), rows = w + nrb + nb, ntile = (rows + 15) >> 4;
#pragma unroll 4
    for (int idx = t; idx < (ngrp + 4) * 4 * W; idx += nt) {      // zero rows behind the last column: the k-loop reads up to 3 groups past it
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
      const int q0 = (wv + NW * i) * 16, q = q0 + lr;
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
      // KD column groups of L in flight per tile while the previous KD are multiplied (4 where the registers allow: the loop
      // waits on loads that come from the L2 / Infinity Cache, 2 waves per SIMD hide little of that)
      constexpr int KD = MT <= 4 ? 4 : 2;
      auto ld = [&](int i, int kg) -> double {
        const int kk = 4 * kg + lq;
        return (kg >= gfirst[i] && kk >= kfirst[i] && kk < nk) ? lp[i][size_t(kk) * lstep[i]] : 0.0;
      };
      double bq[KD][MT];
#pragma unroll
      for (int j = 0; j < KD; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) bq[j][i] = ld(i, j);
      for (int kg = 0; kg < ngrp; kg += KD) {
        double a[KD], nq[KD][MT];
#pragma unroll
        for (int j = 0; j < KD; ++j) a[j] = -T[(4 * (kg + j) + lq) * W + lr];
#pragma unroll
        for (int j = 0; j < KD; ++j)
#pragma unroll
          for (int i = 0; i < MT; ++i) nq[j][i] = ld(i, kg + KD + j);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          if (kg + KD - 1 < gfirst[i]) continue;
#pragma unroll
          for (int j = 0; j < KD; ++j) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[j], bq[j][i], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < KD; ++j)
#pragma unroll
          for (int i = 0; i < MT; ++i) bq[j][i] = nq[j][i];
      }
    }
    IPM_TICK(1);
    if (wv != 0) {
      if (J0 > 0 && nb > 0) corner_update(1, NW - 1);    // the previous block column's, while wave 0 is busy below
    } else {                    // tile 0 holds the diagonal block (rows q < w): its LDL^T and the inverse of L11 by this wave
#if IPM_DIAG_SPLIT
      // the accumulator tile is DiagStep's layout already: lane (lq, lr) holds columns lq + 4 g of row lr
      double R[4], V[4];
      int a_col[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int j = 4 * g + lq;
        R[g] = (lr < w && j <= lr) ? acc[0][g] : (j == lr ? 1.0 : 0.0);
        V[g] = j == lr ? 1.0 : 0.0;
        a_col[g] = j << 2;
      }
      DiagStep<0>::run(R, V, lq, lr, lr << 2, a_col);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int j = 4 * g + lq;
        if (lr < w && j <= lr) Dg[lr * (W + 1) + j] = R[g];
        Mi[lr * W + j] = V[g];
        if (j == lr) invd[lr] = lr < w ? 1.0 / R[g] : 0.0;
      }
#else
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
        // all cross-lane reads of the step first (they are independent), then the arithmetic.  The four groups of 16 lanes hold
        // the same rows, so lane j of the wave serves everybody: scalar reads (v_readlane) instead of LDS-routed shuffles
        const double dk = readlane_d(row[k], k);                   // pivot: row k's own diagonal
        double ajk[W], mkj[W];
#pragma unroll
        for (int j = k + 1; j < W; ++j) ajk[j] = readlane_d(row[k], j);     // a(j, k) before scaling
#pragma unroll
        for (int j = 0; j <= k; ++j) mkj[j] = readlane_d(inv[j], k);        // row k of the inverse so far
        const double lik = lr > k ? row[k] / dk : 0.0;
        // (no `lr >= j` guard: entries right of a lane's diagonal are never read, lanes at or above row k have lik = 0, and the
        // compare and two selects per entry were a quarter of this loop's instructions — the loop is issue-bound)
#pragma unroll
        for (int j = k + 1; j < W; ++j) row[j] = __builtin_fma(-lik, ajk[j], row[j]);
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
#endif
    }
    __syncthreads();
    IPM_TICK(2);
    // panel rows: Y^T = L11^-1 A^T (4 products with the accumulator as B operand), L21^T = D^-1 Y^T
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      if ((wv + NW * i) * 16 >= rows) continue;
      d4 y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int g = 0; g < 4; ++g) y = __builtin_amdgcn_mfma_f64_16x16x4f64(Mi[lr * W + 4 * g + lq], acc[i][g], y, 0, 0, 0);
      const int q = (wv + NW * i) * 16 + lr, r = rrow[i];
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
    IPM_TICK(4);
    J0 = J1;
  }
  if (G.Nb > 0 && nb > 0 && partial != 2) {          // the last block column's corner update
    corner_update(0, NW);
    __syncthreads();
  }
  if (partial == 2) __syncthreads();   // (the corner is in LDS)
  if (partial == 1) {                                            // level 1 of the nested dissection: hand the corner over as it is
    for (int idx = t; idx < nb * nb; idx += nt) {
      const int r = idx / nb, c = idx % nb;
      if (r >= c) K[G.at(G.Nb + r, G.Nb + c)] = C[ct(r, c)];
    }
    if (npos) atomicAdd(&cnt[0], npos);
    if (nneg) atomicAdd(&cnt[1], nneg);
    if (nbad) atomicAdd(&cnt[2], nbad);
    __syncthreads();
    if (t < 3) piv[(size_t(bi) * n_sub + sidx) * 3 + t] = cnt[t];
#ifdef IPM_TIMING
    if (t == 0 && sidx == IPM_TIMING_SUB)
      for (int i = 0; i < 6; ++i) inst[bi].dbg[i] = tc[i];
#endif
    return;
  }
  for (int k = 0; k < nb; ++k) {                            // the corner, unblocked, in LDS
    __syncthreads();
    const double dk = C[ct(k, k)];
    for (int idx = t; idx < nb * nb; idx += nt) {
      const int r = idx / nb, c = idx % nb;
      if (r > k && c > k && c <= r) C[ct(r, c)] -= C[ct(r, k)] / dk * C[ct(c, k)];
    }
    __syncthreads();
    for (int r = k + 1 + t; r < nb; r += nt) C[ct(r, k)] /= dk;
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
        if (i < w2 && c0 + k >= cj) v = __builtin_fma(-C[ct(c0 + i, c0 + k)], x[k], v);
      x[i] = c0 + i < cj ? 0.0 : v;
    }
    // in place: the 16 columns of a sub-block belong to 16 consecutive lanes of one wave, which has read all of them
    // (the loop above) before any lane writes its column back
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < W; ++i)
      if (i < w2 && c0 + i > cj) C[ct(c0 + i, cj)] = x[i];
  }
  __syncthreads();
  for (int idx = t; idx < nb * nb; idx += nt) {
    const int r = idx / nb, c = idx % nb;
    if (r >= c) K[G.at(G.Nb + r, G.Nb + c)] = C[ct(r, c)];
    if (r == c) {
      const double dk = C[ct(r, r)];
      if (dk > 0) ++npos; else if (dk < 0) ++nneg; else ++nbad;
      if (!(fabs(dk) < 1e300)) ++nbad;
    }
  }
  if (npos) atomicAdd(&cnt[0], npos);
  if (nneg) atomicAdd(&cnt[1], nneg);
  if (nbad) atomicAdd(&cnt[2], nbad);
  __syncthreads();
  IPM_TICK(5);
  if (t < 3) piv[(size_t(bi) * n_sub + sidx) * 3 + t] = (partial == 2 ? piv[(size_t(bi) * n_sub + sidx) * 3 + t] : 0) + cnt[t];
#ifdef IPM_TIMING
  if (t == 0 && (IPM_TIMING_SUB < 0 || sidx == IPM_TIMING_SUB))
    for (int i = 0; i < 6; ++i) inst[bi].dbg[i] = tc[i];
#endif
}

// Level 1 of the nested dissection when an interval block is small and (nearly) dense — the quadrotor sweep's are of order 252
// with a half bandwidth of 200: the whole lower triangle lives in the REGISTERS of one workgroup as 16 x 16 accumulator tiles
// (tile (I, K), K <= I, transposed like kkt_factor_kernel's: lane l holds columns (l >> 4) + 4 reg of block K for row l & 15 of
// block I), dealt round robin to 7 waves (at most IPM_DENSE_SLOTS = 22 tiles each: 17 block rows), and the elimination runs
// right-looking: block column J's diagonal tile is factored by the eighth wave, which holds no tiles (the same register LDL^T
// with the inverse alongside; its 100-odd registers would not fit beside 22 accumulator tiles),
// the tiles below it are solved against it (4 matrix products each) and leave L and L D in LDS, every tile to the right takes
// its rank-16 update from there (4 products).  Nothing is read twice: the left-looking kernel streams the 488 KB of such a
// block ~12 times from the L2 / Infinity Cache, a 16-column step every 25 us.  Same products in the same order, so the factors
// are the left-looking kernel's bit for bit.  Partial elimination only (the band part; the border x border corner is handed
// on as the Schur complement).
// EARLY: blocks of more than IPM_DENSE_ROWS block rows (the steps for their first block columns cost the plain kernel 15 %).
// CORNER: a last level — with partial = 0 the corner's block columns are eliminated too, their panels solved by SUBSTITUTION with
// L11 (a lane's 16 steps, each a broadcast among the four lanes of a row) and a true division by d, as kkt_factor_kernel's unblocked
// corner does, instead of the product with the explicit L11^-1: the global border holds the pivots of -delta_c and the
// multipliers of the linkages, and with the inverse there 3 of 28 perturbed starts of the metric problem ended in a failed line search
template <bool EARLY, bool CORNER>
__global__ __launch_bounds__(512, 1) void kkt_factor_dense_kernel(double* Kall, long long kstride, const KktSub* subs, int sub0, int n_here,
                                                                  int n_sub, IpmInst* inst, int* piv, IpmDev D, int assemble, int forward,
                                                                  int partial) {
  typedef double d4 __attribute__((ext_vector_type(4)));
  constexpr int W = IPM_W, NWV = IPM_DENSE_TILE_WAVES, MAXS = IPM_DENSE_SLOTS;
  // the barriers of this kernel order LDS traffic only: __syncthreads() would also wait for the stores of L into the storage
  // (s_waitcnt vmcnt(0)), a trip to the L2 on the critical path of every block column
#define IPM_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
  const int bi = blockIdx.x / n_here, sidx = sub0 + int(blockIdx.x) % n_here, t = threadIdx.x;
  const IpmInst& S = inst[bi];
  if (S.status != 0 || !S.refactor) return;
  const KktSub sub = subs[sidx];
  const KktGeom G = sub.g;
  double* K = Kall + size_t(bi) * kstride + sub.koff;
  const int nbb = (G.Nb + W - 1) / W, nbr = (G.nb + W - 1) / W, NTB = nbb + nbr;
  // the trailing R block rows live in registers; the E block columns before them ("early") go through the storage
  const int E = EARLY ? ipm_dense_early(NTB) : 0, R = NTB - E, ntl = R * (R + 1) / 2;
  const int nbe = (CORNER && !partial) ? NTB : nbb;    // block columns to eliminate: the band part, or (the last level) the corner's as well
  extern __shared__ double lds[];
  double* Dg = lds;                       // 2 x W x (W + 1): the diagonal tile of block column J in copy J & 1 (the next one is handed over
                                          // while the eighth wave still stores the current one)
  constexpr int DGN = W * (W + 1);
  double* Mi = Dg + 2 * DGN;              // W x 18: L11^-1, row-major (rows padded like the panel's, see below)
  double* invd = Mi + W * IPM_DENSE_LDS_ROW;   // W
  // rows of 18 doubles: the 16 lanes that read one column of 16 successive rows then hit 16 different pairs of banks (at 16
  // doubles per row they share two, and the matrix products starve: 18 us per block column instead of 3)
  constexpr int BS = IPM_DENSE_LDS_ROW;
  double* dv = invd + W;                   // W: the pivots d (what the storage holds on the diagonal)
  double* BL = dv + W;                     // NTB x 16 x BS: L of the current block column, by block row
  double* BY = BL + size_t(NTB) * W * BS;  // the same for L D
  double* rsh = BY + size_t(NTB) * W * BS; // forward: this block's right-hand side, NTB x 16
  double* ysh = rsh + size_t(NTB) * W;     // forward: y of the current block column
  const int wv = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, lr = t & 15, lq = (t & 63) >> 4;
  // hand_over: block column J's panel rows of block row J + 1 are in LDS (value J + 1) — all the owner of the next diagonal tile waits for
  __shared__ int hand_over;
  if (t == 0) hand_over = 0;
  // The owner of those rows always reaches its store (no early exit lies before it), so the wait ends; the bound is there so that the
  // grid drains whatever happens — and if it were ever hit the factorisation is NOT delivered: the instance ends with status 5.
  auto wait_hand_over = [&](int value) {
    int spin = 0;
    while (__hip_atomic_load(&hand_over, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < value && ++spin < (1 << 22)) __builtin_amdgcn_s_sleep(1);
    if (spin >= (1 << 22)) inst[bi].status = 5;
  };
  auto row0 = [&](int I) { return I < nbb ? W * I : G.Nb + W * (I - nbb); };
  auto rend = [&](int I) { return I < nbb ? G.Nb : G.Nt; };
  // assemble: the block is built here, not read — the values of its structural slots (a few per cent of the storage) go to LDS
  // (the panel's space, free until the first block column is solved), number 0 being a structural zero; every tile lane then picks
  // its four entries by number (df_map).  The right-hand side entries that go with the diagonal slots are written on the way.
  double* vals = BL;
  unsigned long long mp[MAXS];     // a tile wave's lanes: the numbers of the entries they hold, on their way while the values are fetched
  if (assemble) {
    if (wv != NWV) {
#pragma unroll
      for (int s = 0; s < MAXS; ++s) mp[s] = D.df_map[(size_t(sidx) * D.df_tiles + wv + NWV * s) * 64 + lane];
    }
    const int ea = D.df_ptr[3 * sidx], eb = D.df_ptr[3 * sidx + 1], ec = D.df_ptr[3 * sidx + 2], ed = D.df_ptr[3 * sidx + 3];
    const SlotValue value(D, bi);
    if (t == 0) vals[0] = 0.0;
    const double* jac = D.jac + size_t(bi) * D.sv;
#pragma unroll 4
    for (int e = ea + t; e < eb; e += 512) vals[1 + e - ea] = jac[D.df_ki[e] & 0x0fffffff];
    const bool with_h = value.mode == 0 && !D.lb_on;
    const double* hess = D.hess + size_t(bi) * D.nnz_h;
#pragma unroll 2
    for (int e = eb + t; e < ec; e += 512) {
      double a = 0.0;
      if (with_h) {
        const int hgi = D.df_ki[e] & 0x0fffffff;
        for (int j = D.hg_ptr[hgi]; j < D.hg_ptr[hgi + 1]; ++j) a += hess[D.hg_src[j]];
      }
      vals[1 + e - ea] = 0.0 + a;
    }
    for (int e = ec + t; e < ed; e += 512) vals[1 + e - ea] = value(D.df_ki, D.df_hg, e);
    __syncthreads();
  }
  // forward: the first half of the substitution that follows an accepted factorisation (kkt_solve_kernel's phase 1: L y = r over the
  // band blocks, the border work space takes -L_border y) runs along with the elimination — the panel of a block column is in
  // LDS anyway, and the factor is not streamed from HBM a second time for it.  Same sums in the same order as that kernel's.
  double* rg = D.rhs + size_t(bi) * D.Nt + sub.roff;
  if (forward) {
    for (int i = t; i < G.Nt; i += 512) {
      const int I = i < G.Nb ? i / W : nbb + (i - G.Nb) / W;
      rsh[I * W + (i - row0(I))] = rg[i];
    }
  }
  if (wv == NWV) {
    // ---------------- the eighth wave holds no tiles: it factors the diagonal blocks (its registers are free for that) ----------------
    __builtin_amdgcn_s_setprio(3);   // the chain of diagonal blocks is the critical path: this wave goes first on its SIMD
    int npos = 0, nneg = 0, nbad = 0;
#ifdef IPM_TIMING
    long long tq[4] = {0, 0, 0, 0}, tp = wall_clock64();
#define IPM_DTICK(i) do { const long long n_ = wall_clock64(); tq[i] += n_ - tp; tp = n_; } while (0)
#else
#define IPM_DTICK(i)
#endif
    for (int J = 0; J < nbe; ++J) {
      const int J0 = row0(J), w = min(W, rend(J) - J0);
      double* DgJ = Dg + (J & 1) * DGN;
      IPM_LDS_BARRIER();          // B1: the owner of tile (J, J) has put it into its copy of Dg — and, from J = 1 on, block column J - 1's panel is in LDS
      IPM_DTICK(0);
#if IPM_DIAG_SPLIT
      {   // kkt_factor_kernel's elimination, entry for entry, over all 64 lanes (DiagStep)
        double R[4], V[4];
        int a_col[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int j = 4 * g + lq;
          R[g] = (lr < w && j <= lr) ? DgJ[lr * (W + 1) + j] : (j == lr ? 1.0 : 0.0);
          V[g] = j == lr ? 1.0 : 0.0;
          a_col[g] = j << 2;
        }
        DiagStep<0>::run(R, V, lq, lr, lr << 2, a_col);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int j = 4 * g + lq;
          if (lr < w && j <= lr) DgJ[lr * (W + 1) + j] = R[g];
          Mi[lr * IPM_DENSE_LDS_ROW + j] = V[g];
          if (j == lr) {
            invd[lr] = lr < w ? 1.0 / R[g] : 0.0;
            dv[lr] = lr < w ? R[g] : 0.0;
          }
        }
      }
#else
      double row[W], inv[W];    // lane (l & 15) = row of the block and of L11^-1 (kkt_factor_kernel's elimination, word for word)
#pragma unroll
      for (int c = 0; c < W; ++c) {
        row[c] = (lr < w && c <= lr) ? DgJ[lr * (W + 1) + c] : (c == lr ? 1.0 : 0.0);
        inv[c] = c == lr ? 1.0 : 0.0;
      }
#pragma unroll
      for (int k = 0; k < W; ++k) {
        const double dk = readlane_d(row[k], k);
        double ajk[W], mkj[W];
#pragma unroll
        for (int j = k + 1; j < W; ++j) ajk[j] = readlane_d(row[k], j);
#pragma unroll
        for (int j = 0; j <= k; ++j) mkj[j] = readlane_d(inv[j], k);
        // (a reciprocal estimate with two Newton steps instead of the division: 90.5 -> 86.6 us of diagonal-block time per
        // interval block, not worth leaving the left-looking kernel's arithmetic)
        const double lik = lr > k ? row[k] / dk : 0.0;
        // (no `lr >= j` guard as in kkt_factor_kernel: entries right of a lane's diagonal are never read, and the compare and
        // the two selects per entry are a quarter of this loop's instructions — it is issue-bound, 800 cycles per step)
#pragma unroll
        for (int j = k + 1; j < W; ++j) row[j] = __builtin_fma(-lik, ajk[j], row[j]);
#pragma unroll
        for (int j = 0; j <= k; ++j) inv[j] = __builtin_fma(-lik, mkj[j], inv[j]);
        if (lr > k) row[k] = lik;
      }
      if (lq == 0) {
#pragma unroll
        for (int c = 0; c < W; ++c) {
          if (lr < w && c <= lr) DgJ[lr * (W + 1) + c] = row[c];
          Mi[lr * IPM_DENSE_LDS_ROW + c] = inv[c];
        }
        invd[lr] = lr < w ? 1.0 / row[lr] : 0.0;
        dv[lr] = lr < w ? row[lr] : 0.0;
      }
#endif
      IPM_DTICK(1);
      IPM_LDS_BARRIER();          // B2: Dg, Mi, invd are there
      if (forward) {              // y = L11^-1 r of this block's unknowns (r is final: the tile waves added the last panel's share before B2),
        double y = lr < w ? rsh[J * W + lr] : 0.0;   // while they solve the panel
#pragma unroll
        for (int k = 0; k < W; ++k) {
          const double zk = k < w ? rsh[J * W + k] : 0.0;
          if (k < lr && lr < w) y = __builtin_fma(Mi[lr * IPM_DENSE_LDS_ROW + k], zk, y);
        }
        if (lq == 0) {
          ysh[lr] = y;
          if (lr < w) rg[J0 + lr] = y;
        }
      }
      // the diagonal block is stored as d on the diagonal and L11^-1 below it (what the solves use)
      for (int idx = lane; idx < W * W; idx += 64) {
        const int di = idx / W, dj = idx % W;
        if (di < w && dj <= di) K[G.at(J0 + di, J0 + dj)] = di == dj ? DgJ[di * (W + 1) + dj] : Mi[di * IPM_DENSE_LDS_ROW + dj];
      }
      if (lane < w) {
        const double dk = DgJ[lane * (W + 1) + lane];
        if (dk > 0) ++npos; else if (dk < 0) ++nneg; else ++nbad;
        if (!(fabs(dk) < 1e300)) ++nbad;
      }
      IPM_DTICK(2);
    }
    IPM_LDS_BARRIER();            // the last block column's panel is in LDS (the tile waves' last meeting point)
#ifdef IPM_TIMING
    if (lane == 0 && sidx == 0) { inst[bi].dbg[6] = tq[0] + tq[2]; inst[bi].dbg[7] = tq[1]; }   // waiting for the tile waves | factoring
#endif
    for (int o = 32; o; o >>= 1) { npos += __shfl_xor(npos, o); nneg += __shfl_xor(nneg, o); nbad += __shfl_xor(nbad, o); }
    if (lane == 0) {
      int* pv = piv + (size_t(bi) * n_sub + sidx) * 3;
      pv[0] = npos; pv[1] = nneg; pv[2] = nbad;
    }
    return;
  }
  // ---------------- tile waves ----------------
  // Tiles are numbered column by column (tile (I, K): K NTB - K (K - 1) / 2 + I - K) and dealt round robin, slot s of wave wv
  // holds tile wv + 7 s: the tiles a step touches are then a RANGE of slots — those of block column J for the panel solve,
  // everything from the first tile of column J + 1 on for the update — entered through a switch and walked without a branch
  // per tile, so that the LDS reads and matrix products of successive tiles overlap.
#define IPM_REP22(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15) M(16) M(17) M(18) M(19) M(20) M(21)
  static_assert(MAXS == 22, "IPM_REP22");
  auto colstart = [&](int Kr) { return Kr * R - Kr * (Kr - 1) / 2; };   // first resident tile of block column E + Kr
  // a tile of an early block column: from / to the storage, in the register tiles' layout (tile (I, Kb) belongs to wave I % 7)
  auto t_load = [&](int I, int Kb) {
    d4 a;
    const int r = row0(I) + lr;
    const bool rv = r < rend(I), border = r >= G.Nb;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int cc = row0(Kb) + lq + 4 * g;
      a[g] = (rv && cc < rend(Kb) && cc <= r && (border || r - cc <= G.b)) ? K[G.at(r, cc)] : 0.0;
    }
    return a;
  };
  auto t_store = [&](const d4& a, int I, int Kb) {
    const int r = row0(I) + lr;
    const bool rv = r < rend(I), border = r >= G.Nb;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int cc = row0(Kb) + lq + 4 * g;
      if (rv && cc < rend(Kb) && cc <= r && (border || r - cc <= G.b)) K[G.at(r, cc)] = a[g];
    }
  };
  auto t_update = [&](d4 a, int I, int Kb) {       // A(I, Kb) -= L(I, J) D L(Kb, J)^T from the panel in LDS (IPM_UPD_BODY's products)
    const double* bl = BL + (I * W + lr) * BS + lq;
    const double* by = BY + (Kb * W + lr) * BS + lq;
#pragma unroll
    for (int g = 0; g < 4; ++g) a = __builtin_amdgcn_mfma_f64_16x16x4f64(-by[4 * g], bl[4 * g], a, 0, 0, 0);
    return a;
  };
  auto t_put = [&](const d4& a, int wd, double* dg) {   // a diagonal tile of width wd: its lower triangle to Dg
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int c = lq + 4 * g;
      if (lr < wd && c <= lr) dg[lr * (W + 1) + c] = a[g];
    }
  };
  if (assemble && E > 0) {       // the early tiles are built in the storage (the values' LDS space is the panel's)
    for (int Kb = 0; Kb < E; ++Kb)
      for (int I = Kb; I < NTB; ++I) {
        if (I % NWV != wv) continue;
        const unsigned long long m = D.df_map[(size_t(sidx) * D.df_tiles + ipm_dense_tile(NTB, I, Kb)) * 64 + lane];
        d4 a;
#pragma unroll
        for (int g = 0; g < 4; ++g) a[g] = vals[(m >> (16 * g)) & 0xffff];
        t_store(a, I, Kb);
      }
  }
  int sIK[MAXS];                // block row << 8 | block column of the slot's tile (wave-uniform); unused slots: tile (0, 0), never stored
  d4 acc[MAXS];
#pragma unroll
  for (int s = 0; s < MAXS; ++s) {
    const int tl = wv + NWV * s;
    int Kb = 0;
    while (Kb + 1 < R && colstart(Kb + 1) <= tl) ++Kb;
    const bool have = tl < ntl;
    const int I = have ? E + Kb + (tl - colstart(Kb)) : 0;
    Kb = have ? E + Kb : 0;
    sIK[s] = __builtin_amdgcn_readfirstlane(I << 8 | Kb);
    if (assemble) {
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[s][g] = vals[(mp[s] >> (16 * g)) & 0xffff];
      continue;
    }
    const int r = row0(I) + lr;
    const bool rv = have && r < rend(I), border = r >= G.Nb;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int cc = row0(Kb) + lq + 4 * g;
      const bool ok = rv && cc < rend(Kb) && cc <= r && (border || r - cc <= G.b);
      acc[s][g] = ok ? K[G.at(r, cc)] : 0.0;
    }
  }
  auto first_slot_at = [&](int tl) { return tl <= wv ? 0 : (tl - wv + NWV - 1) / NWV; };   // first slot of this wave with tile number >= tl
  // slot s takes its rank-16 update from the panel in LDS: A(I, K) -= L(I, J) D L(K, J)^T
#define IPM_UPD_BODY(s) {                                                                                                    \
    int ik = sIK[s];                                                                                                         \
    asm volatile("" : "+s"(ik));   /* keeps the 44 LDS addresses from being hoisted out of the J loop into registers */      \
    const double* bl = BL + ((ik >> 8) * W + lr) * BS + lq;                                                                  \
    const double* by = BY + ((ik & 255) * W + lr) * BS + lq;                                                                 \
    _Pragma("unroll") for (int g = 0; g < 4; ++g) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(-by[4 * g], bl[4 * g], acc[s], 0, 0, 0); \
  }
  // slot s is a diagonal tile of width wd: its lower triangle goes to Dg for the eighth wave
#define IPM_PUT_BODY(s, wd, dg) { _Pragma("unroll") for (int g = 0; g < 4; ++g) { const int c = lq + 4 * g; if (lr < (wd) && c <= lr) (dg)[lr * (W + 1) + c] = acc[s][g]; } }
  if (wv == 0) {
    if (E == 0) IPM_PUT_BODY(0, min(W, rend(0)), Dg)    // tile (0, 0) is tile number 0: slot 0 of wave 0
    else t_put(t_load(0, 0), min(W, rend(0)), Dg);
  }
  // early block column J: this wave's tiles (I, J), I > J, in ebuf (I = first such I + 7 q) for its panel only — kept across the
  // resident tiles' update they pushed those out of the registers —, tile (J + 1, J + 1) in edg with the wave that owns it, fetched a step ahead
  constexpr int EQ = (IPM_DENSE_ROWS + IPM_DENSE_EARLY - 1 + NWV - 1) / NWV;
  d4 ebuf[EQ], edg;
  auto first_own = [&](int from) { return from + (wv - from % NWV + NWV) % NWV; };   // first I >= from with I % 7 == wv
  if (E > 0) {
    const int f0 = first_own(1);
#pragma unroll
    for (int q = 0; q < EQ; ++q)
      if (f0 + NWV * q < NTB) ebuf[q] = t_load(f0 + NWV * q, 0);
    if (E > 1 && 1 % NWV == wv) edg = t_load(1, 1);
  }
  IPM_LDS_BARRIER();              // B1 of block column 0
  IPM_LDS_BARRIER();              // B2: its diagonal block is factored
#ifdef IPM_TIMING
  long long tw[6] = {0, 0, 0, 0, 0, 0}, twp = wall_clock64();
#ifdef IPM_TIMING_EARLY   // the phases of the EARLY block columns in the record's places, everything else in "early"
#define IPM_ETICK(i) do { const long long n_ = wall_clock64(); tw[i] += n_ - twp; twp = n_; } while (0)
#define IPM_TTICK(i) do { const long long n_ = wall_clock64(); tw[5] += n_ - twp; twp = n_; } while (0)
#else
#define IPM_TTICK(i) do { const long long n_ = wall_clock64(); tw[i] += n_ - twp; twp = n_; } while (0)
#define IPM_ETICK(i) IPM_TTICK(5)
#endif
#else
#define IPM_TTICK(i)
#define IPM_ETICK(i)
#endif
  // forward: r(row) -= L(row, J) y for every row below block column J, thread p owns row p of the tile rows.  Off the chain of
  // diagonal blocks: after the next diagonal tile has been handed over (the panel and y stay in LDS until the next B2)
  auto forward_share = [&](int J) {
    const int I = t >> 4;
    if (I > J && I < NTB && row0(I) + lr < rend(I)) {
      const double* bl = BL + (size_t(I) * W + lr) * BS;
      double a = 0.0;
#pragma unroll
      for (int c = 0; c < W; ++c) a = __builtin_fma(bl[c], ysh[c], a);
      rsh[I * W + lr] -= a;
    }
  };
  // a corner tile's panel solve by substitution: Y(r, c) = A(r, c) - sum_{k < c} Y(r, k) L11(c, k), L11 below the diagonal of Dg
  auto corner_subst = [&](const d4& a, int J) {
    const double* DgJ = Dg + (J & 1) * DGN;
    d4 y = a;
#pragma unroll
    for (int c = 0; c < W - 1; ++c) {
      const double yc = __shfl(y[c >> 2], lr + 16 * (c & 3));   // Y(r, c) is final: from the lane of this row that holds column c
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int j = lq + 4 * g;
        if (j > c) y[g] = __builtin_fma(-yc, DgJ[j * (W + 1) + c], y[g]);
      }
    }
    return y;
  };
  for (int J = 0; J < nbe; ++J) {
    const int J0 = row0(J), w = min(W, rend(J) - J0);
    int s0_early = 0;
    if (EARLY && J < E) {
      // ---- an early block column: the same steps with its tiles (and those of the early columns to its right) taken from the
      // storage, each by the wave that owns it; the resident tiles take their update as always
      IPM_ETICK(4);
      const int f0 = first_own(J + 1);
      if (J > 0) {              // this column's tiles: fetched now (they have the updates of the columns before J - 1), updated with
#pragma unroll                  // column J - 1's panel, which is still in LDS
        for (int q = 0; q < EQ; ++q)
          if (f0 + NWV * q < NTB) ebuf[q] = t_load(f0 + NWV * q, J);
#pragma unroll
        for (int q = 0; q < EQ; ++q)
          if (f0 + NWV * q < NTB) ebuf[q] = t_update(ebuf[q], f0 + NWV * q, J);
      }
#pragma unroll
      for (int q = 0; q < EQ; ++q) {
        const int I = f0 + NWV * q;
        if (I >= NTB) break;
        d4 y = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int g = 0; g < 4; ++g) y = __builtin_amdgcn_mfma_f64_16x16x4f64(Mi[lr * IPM_DENSE_LDS_ROW + 4 * g + lq], ebuf[q][g], y, 0, 0, 0);
        const int r = row0(I) + lr;
        const bool rv = r < rend(I), border = r >= G.Nb;
        const int kstep = border ? G.CS : G.CS - 1;
        double* kp = K + (size_t(J0) * G.CS + (border ? G.b + 1 + r - G.Nb : r - J0)) + lq * kstep;
        const int lo_ = (I * W + lr) * BS + lq;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c = lq + 4 * g;
          const double l = y[g] * invd[c];
          const bool ok = rv && c < w;
          if (ok && (border || r - (J0 + c) <= G.b)) kp[4 * g * kstep] = l;
          BL[lo_ + 4 * g] = ok ? l : 0.0;
          BY[lo_ + 4 * g] = ok ? (border ? y[g] : l * dv[c]) : 0.0;
        }
        if (I == J + 1) __hip_atomic_store(&hand_over, J + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      IPM_ETICK(0);
      int s0 = 0;
      if (J + 1 < nbe) {        // the next diagonal tile, before the workgroup meets
        const int w1 = min(W, rend(J + 1) - row0(J + 1));
        double* DgN = Dg + ((J + 1) & 1) * DGN;
        if (J + 1 < E) {
          if ((J + 1) % NWV == wv) t_put(t_update(edg, J + 1, J + 1), w1, DgN);   // (this wave wrote the panel's rows it needs)
        } else if (wv == 0) {   // the first resident tile: slot 0 of wave 0
          wait_hand_over(J + 1);
          IPM_UPD_BODY(0) IPM_PUT_BODY(0, w1, DgN)
          s0 = 1;
        }
      }
      s0_early = s0;
    }
    const int cs = colstart(max(J - E, 0)), cs1 = colstart(max(J - E, 0) + 1);
    if (!(EARLY && J < E)) {
    IPM_TTICK(4);               // waiting at B2
    // the tiles below the diagonal one: Y^T = L11^-1 A^T, L^T = D^-1 Y^T; both go to LDS for the updates, L to the storage
    {
      const int sa = first_slot_at(cs + 1), sb = min(MAXS, first_slot_at(cs1));   // slots [sa, sb)
      const bool first_owner = (cs + 1) % NWV == wv;   // this wave's first tile of the column is (J + 1, J): the rows the next diagonal tile needs
      if (CORNER && J >= nbb) {
        // a corner block column: one copy of the substitution code for all slots (inside the switch it would be there 22 times)
        for (int sl = sa; sl < sb; ++sl) {
          d4 a = {0.0, 0.0, 0.0, 0.0};
          switch (sl) {
#define IPM_PICK(s) case s: a = acc[s]; break;
            IPM_REP22(IPM_PICK)
#undef IPM_PICK
            default: break;
          }
          const d4 y = corner_subst(a, J);
          const int I = J + (wv + NWV * sl - cs), r = row0(I) + lr;
          const bool rv = r < rend(I);
          double* kp = K + (size_t(J0) * G.CS + (G.b + 1 + r - G.Nb)) + lq * G.CS;      // (every row below a corner column is a border row)
          const int lo_ = (I * W + lr) * BS + lq;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int c = lq + 4 * g;
            const bool ok = rv && c < w;
            const double l = ok ? y[g] / dv[c] : 0.0;
            if (ok) kp[4 * g * G.CS] = l;
            BL[lo_ + 4 * g] = l;
            BY[lo_ + 4 * g] = ok ? y[g] : 0.0;
          }
          if (sl == sa && first_owner) __hip_atomic_store(&hand_over, J + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      } else
      switch (sa) {
#define IPM_PANEL(s) case s: if (s < sb) {                                                                                   \
          d4 y = {0.0, 0.0, 0.0, 0.0};                                                                                       \
          _Pragma("unroll") for (int g = 0; g < 4; ++g) y = __builtin_amdgcn_mfma_f64_16x16x4f64(Mi[lr * IPM_DENSE_LDS_ROW + 4 * g + lq], acc[s][g], y, 0, 0, 0); \
          int ik = sIK[s];                                                                                                   \
          asm volatile("" : "+s"(ik));                                                                                       \
          const int I = ik >> 8, r = row0(I) + lr;                                                                           \
          const bool rv = r < rend(I), border = r >= G.Nb;                                                                   \
          /* the four entries of a lane sit 4 columns apart: one 64-bit address (row r, column J0 + lq) and a 32-bit step — a   \
             per-entry G.at() is a 64-bit vector multiply each, and the stores' address arithmetic was 60 % of the panel's time */ \
          const int kstep = border ? G.CS : G.CS - 1;                                                                        \
          double* kp = K + (size_t(J0) * G.CS + (border ? G.b + 1 + r - G.Nb : r - J0)) + lq * kstep;                         \
          const int lo_ = (I * W + lr) * BS + lq;                                                                            \
          _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                                    \
            const int c = lq + 4 * g;                                                                                        \
            const double l = y[g] * invd[c];                                                                                 \
            const bool ok = rv && c < w;                                                                                     \
            if (ok && (border || r - (J0 + c) <= G.b)) kp[4 * g * kstep] = l;                                                \
            BL[lo_ + 4 * g] = ok ? l : 0.0;                                                                                  \
            /* L D as kkt_factor_kernel forms it: the product of the stored l and d for a band row (its T), y itself for a   \
               border row (its BY) — the two differ in the last bit, and Delta-III's path is sensitive to that */            \
            BY[lo_ + 4 * g] = ok ? (border ? y[g] : l * dv[c]) : 0.0;                                                        \
          }                                                                                                                  \
          if (s == sa && first_owner) __hip_atomic_store(&hand_over, J + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); \
        }
        IPM_REP22(IPM_PANEL)
#undef IPM_PANEL
        default: break;
      }
    }
    IPM_TTICK(0);               // panel
    }
    // every tile to the right takes its update; the next diagonal tile first and BEFORE the workgroup meets: its owner needs the
    // panel's rows of block row J + 1 only, which their owner announces (hand_over) after its first tile.  One barrier then says
    // both "the panel is in LDS" and "the next diagonal tile is in its copy of Dg" (they were two, with this update between them)
    int s0 = (EARLY && J < E) ? s0_early : first_slot_at(cs1);
    if (!(EARLY && J < E) && J + 1 < nbe && cs1 % NWV == wv) {
      const int w1 = min(W, rend(J + 1) - row0(J + 1));
      double* DgN = Dg + ((J + 1) & 1) * DGN;
      wait_hand_over(J + 1);
      switch (s0) {
#define IPM_NEXT(s) case s: IPM_UPD_BODY(s) IPM_PUT_BODY(s, w1, DgN) break;
        IPM_REP22(IPM_NEXT)
#undef IPM_NEXT
        default: break;
      }
      ++s0;
    }
    IPM_TTICK(2);               // next diagonal tile
    IPM_LDS_BARRIER();            // the panel is in LDS; B1 of block column J + 1
    IPM_TTICK(1);               // waiting there
    if (forward) forward_share(J);
    switch (s0) {
#define IPM_UPD(s) case s: IPM_UPD_BODY(s) if (s & 1) __builtin_amdgcn_sched_barrier(0);   /* two tiles' loads and products may interleave, not all 22 (registers) */
      IPM_REP22(IPM_UPD)
#undef IPM_UPD
      default: break;
    }
    if (EARLY && J < E) {
      for (int Kb = J + 2; Kb < E; ++Kb)      // early columns further right: through the storage
        for (int I = first_own(Kb); I < NTB; I += NWV) t_store(t_update(t_load(I, Kb), I, Kb), I, Kb);
      if (J + 2 < E && (J + 2) % NWV == wv) edg = t_load(J + 2, J + 2);
    }
    IPM_TTICK(3);               // update
    if (J + 1 < nbe) IPM_LDS_BARRIER();   // B2 of block column J + 1
  }
#ifdef IPM_TIMING
  // (build with -DIPM_TIMING_SUB=<out of range> so that the left-looking kernel of the last level leaves these alone:
  //  wave 0's panel | wait at B3 | next diagonal tile + B1 | update | wait at B2, 100 MHz ticks per interval block)
  if (t == 0 && sidx == 0) { inst[bi].dbg[0] = tw[0]; inst[bi].dbg[1] = tw[1]; inst[bi].dbg[2] = tw[2]; inst[bi].dbg[3] = tw[3]; inst[bi].dbg[4] = tw[4]; inst[bi].dbg[5] = tw[5]; }
#endif
#undef IPM_UPD_BODY
#undef IPM_PUT_BODY
  if (forward) {   // the border work space (each row by the thread that kept it)
    const int I = t >> 4;
    if (I >= nbb && I < NTB && row0(I) + lr < rend(I)) rg[row0(I) + lr] = rsh[I * W + lr];
  }
  // the Schur complement of the corner, unfactored, back into the corner's storage
#pragma unroll
  for (int s = 0; s < MAXS; ++s) {
    if (partial && (sIK[s] & 255) >= nbb && wv + NWV * s < ntl) {
      const int r = row0(sIK[s] >> 8) + lr;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int cc = row0(sIK[s] & 255) + lq + 4 * g;
        if (r < G.Nt && cc < G.Nt && cc <= r) K[G.at(r, cc)] = acc[s][g];
      }
    }
  }
#undef IPM_REP22
#undef IPM_LDS_BARRIER
}
size_t kkt_factor_dense_lds_bytes(int block_rows) {
  return (2 * size_t(IPM_W) * (IPM_W + 1) + size_t(IPM_W) * IPM_DENSE_LDS_ROW + 2 * IPM_W + 2 * size_t(block_rows) * IPM_W * IPM_DENSE_LDS_ROW +
          size_t(block_rows) * IPM_W + IPM_W) * sizeof(double);
}
int kkt_factor_dense_max_block_rows() {   // 7 waves x IPM_DENSE_SLOTS tiles hold the lower triangle of IPM_DENSE_ROWS block rows; the early columns on top
  return IPM_DENSE_ROWS + IPM_DENSE_EARLY;
}
hipError_t kkt_factor_dense_prepare(size_t lds_bytes) {
  if (lds_bytes <= 48 * 1024) return hipSuccess;
  const int plain = int(std::min(lds_bytes, kkt_factor_dense_lds_bytes(IPM_DENSE_ROWS)));
  hipError_t er = hipFuncSetAttribute(reinterpret_cast<const void*>(kkt_factor_dense_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, plain);
  if (er == hipSuccess)
    er = hipFuncSetAttribute(reinterpret_cast<const void*>(kkt_factor_dense_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, plain);
  if (er != hipSuccess || lds_bytes <= kkt_factor_dense_lds_bytes(IPM_DENSE_ROWS)) return er;
  er = hipFuncSetAttribute(reinterpret_cast<const void*>(kkt_factor_dense_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes));
  if (er != hipSuccess) return er;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kkt_factor_dense_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes));
}
// (the LDS size says how many block rows a level's largest block has)
static bool dense_early(size_t lds_bytes) { return lds_bytes > kkt_factor_dense_lds_bytes(IPM_DENSE_ROWS); }

// L y = r, then x = L^-T D^-1 y, in place in rhs: one workgroup per instance, IPM_W columns per step.  The diagonal
// blocks hold L11^-1, so a step's own 16 unknowns are 16 parallel dot products.  The right-hand side lives in LDS when it
// fits (RL); the diagonal block and each thread's panel row of the NEXT step are fetched while the current one is worked.
template <bool RL, int PF = 1>
__global__ __launch_bounds__(256) void kkt_solve_kernel(const double* Kall, long long kstride, const KktSub* subs, int sub0, int n_here,
                                                        const IpmInst* inst, double* rhs_all, long long rhs_stride, int check_status,
                                                        int phase, int kmod) {
  // phase 0: forward and backward over all blocks; nested dissection level 1: phase 1 = forward over the band blocks only
  // (the border work space receives -L_border y, this interval's contribution to the separator system's right-hand side),
  // phase 2 = backward over the band blocks only (the work space then holds the separator / border solution)
  constexpr int W = IPM_W;
  // with the right-hand side in LDS the barriers order LDS traffic only: __syncthreads() would also wait for the factor entries
  // fetched for the NEXT step (s_waitcnt vmcnt(0)), a trip to the L2 / HBM on the chain of every step
#define SOLVE_BARRIER() do { if (RL) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); else __syncthreads(); } while (0)
  // several right-hand sides per instance (IpmDev::rhs_mult): right-hand side bi belongs to instance bi % kmod
  const int bi = blockIdx.x / n_here, bk = bi % kmod, t = threadIdx.x, nt = blockDim.x;
  if (check_status && (inst[bk].status != 0 || (check_status == 2 && !inst[bk].soc_req))) return;
  const KktSub sub = subs[sub0 + int(blockIdx.x) % n_here];
  const KktGeom G = sub.g;
  const double* K = Kall + size_t(bk) * kstride + sub.koff;
  double* rg = rhs_all + size_t(bi) * rhs_stride + sub.roff;
  extern __shared__ double rsh[];
  double* r = RL ? rsh : rg;
  __shared__ double Dg[W * (W + 1)], ys[W], zs[W], red[4][W];     // Dg: d on the diagonal, L11^-1 below it
  const int di = t / W, dj = t % W;
  const int nbb = (G.Nb + W - 1) / W, ncb = (G.nb + W - 1) / W, nblk = nbb + ncb;
  if (RL) {
    for (int i = t; i < G.Nt; i += nt) rsh[i] = rg[i];
    SOLVE_BARRIER();
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
    // one 64-bit address (row, column J0) and a 32-bit step per column (a G.at() per entry is a 64-bit multiply each: most of a step's instructions)
    const bool brd = row >= G.Nb;
    const int kstep = brd ? G.CS : G.CS - 1;
    const double* kp = K + (size_t(B.J0) * G.CS + (brd ? G.b + 1 + row - G.Nb : max(row - B.J0, 0)));
#pragma unroll
    for (int c = 0; c < W; ++c)
      l[c] = (row >= 0 && c < B.w && (brd || row - (B.J0 + c) <= G.b)) ? kp[c * kstep] : 0.0;
  };
  // PF = 2 (launches of few workgroups, where occupancy is no concern): two steps' shares on their way — a step is a few hundred cycles of
  // LDS work between barriers, a trip to the L2 / HBM takes longer
  double dg0, l0[W], dg1, l1[W];
  const int fwd_end = phase == 1 ? nbb : (phase == 2 ? 0 : nblk), bwd_begin = phase == 2 ? nbb : (phase == 1 ? 0 : nblk);
  auto forward_step = [&](int blk, double& dg, double (&l)[W]) {
    const Blk B = blk_of(blk);
    if (di < W && dj <= di) Dg[di * (W + 1) + dj] = dg;
    if (t < W) zs[t] = t < B.w ? r[B.J0 + t] : 0.0;
    SOLVE_BARRIER();
    if (t < W) {                // y = L11^-1 r: 16 lanes, one row each
      double y = zs[t];
#pragma unroll
      for (int k = 0; k < W; ++k)
        if (k < t && t < B.w) y = __builtin_fma(Dg[t * (W + 1) + k], zs[k], y);
      ys[t] = y;
      if (t < B.w) r[B.J0 + t] = y;
    }
    SOLVE_BARRIER();
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
    fetch(blk + PF, dg, l);     // in flight across the barrier and the next PF - 1 steps
    SOLVE_BARRIER();
  };
  if (fwd_end > 0) { fetch(0, dg0, l0); if (PF == 2) fetch(1 < fwd_end ? 1 : -1, dg1, l1); }
  for (int blk = 0; blk < fwd_end; blk += PF) {
    forward_step(blk, dg0, l0);
    if (PF == 2 && blk + 1 < fwd_end) forward_step(blk + 1, dg1, l1);
  }
  auto backward_step = [&](int blk, double& dg, double (&l)[W]) {
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
    fetch(blk - PF, dg, l);     // in flight across the reduction, the diagonal solve and the next PF - 1 steps
    {   // the 16 sums over the wave, each by the same tree as `for (o = 32; o; o >>= 1) v += shfl_down(v, o)` (lane l + lane l + o:
        // the same pairs, a + b for b + a at most), but the columns are dealt out while the lanes fold: 8 + 4 + 2 + 1 + 1 + 1
        // exchanges instead of 16 x 6 — column c's sum ends in lane 4 c.  (The LDS pipe, which carries the exchanges, bounded the
        // backward pass when 12 right-hand sides of the limited-memory update run side by side.)
      const int ln = t & 63;
      double q8[8], q4[4], q2[2], u;
      const bool h32 = ln & 32, h16 = ln & 16, h8 = ln & 8, h4 = ln & 4;
#pragma unroll
      for (int i = 0; i < 8; ++i) q8[i] = (h32 ? p[i + 8] : p[i]) + __shfl_xor(h32 ? p[i] : p[i + 8], 32);
#pragma unroll
      for (int i = 0; i < 4; ++i) q4[i] = (h16 ? q8[i + 4] : q8[i]) + __shfl_xor(h16 ? q8[i] : q8[i + 4], 16);
#pragma unroll
      for (int i = 0; i < 2; ++i) q2[i] = (h8 ? q4[i + 2] : q4[i]) + __shfl_xor(h8 ? q4[i] : q4[i + 2], 8);
      u = (h4 ? q2[1] : q2[0]) + __shfl_xor(h4 ? q2[0] : q2[1], 4);
      u += __shfl_xor(u, 2);
      u += __shfl_xor(u, 1);
      if ((ln & 3) == 0) red[t >> 6][ln >> 2] = u;
    }
    SOLVE_BARRIER();
    if (t < W) zs[t] = t < B.w ? r[B.J0 + t] / Dg[t * (W + 1) + t] - (red[0][t] + red[1][t] + red[2][t] + red[3][t]) : 0.0;
    SOLVE_BARRIER();
    if (t < B.w) {              // x = L11^-T z
      double x = zs[t];
#pragma unroll
      for (int k = 0; k < W; ++k)
        if (k > t && k < B.w) x = __builtin_fma(Dg[k * (W + 1) + t], zs[k], x);
      r[B.J0 + t] = x;
    }
    SOLVE_BARRIER();
  };
  if (bwd_begin > 0) { fetch(bwd_begin - 1, dg0, l0); if (PF == 2) fetch(bwd_begin - 2, dg1, l1); }
  for (int blk = bwd_begin - 1; blk >= 0; blk -= PF) {
    backward_step(blk, dg0, l0);
    if (PF == 2 && blk - 1 >= 0) backward_step(blk - 1, dg1, l1);
  }
  if (RL)
    for (int i = t; i < G.Nt; i += nt) rg[i] = rsh[i];
#undef SOLVE_BARRIER
}

// ------------------------------------------------------------------------------------------------ inertia correction
// Algorithm IC: the factorisation is accepted when D has exactly nv positive entries (and no zero / NaN pivot)
__global__ __launch_bounds__(256) void ipm_inertia_kernel(IpmDev D) {   // a wave per instance: its lanes add up the sub-problems' pivot counts
  const int bi = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (bi >= D.B) return;
  IpmInst& S = D.inst[bi];
  if (S.status != 0 || !S.refactor) return;
  const IpmOpts& o = D.o;
  int np = 0, nn = 0, nz = 0;
  for (int s2 = lane; s2 < D.n_sub; s2 += 64) {   // pivot signs of all the sub-problems of this instance (one without dissection)
    const int* q = D.piv + (size_t(bi) * D.n_sub + s2) * 3;
    np += q[0]; nn += q[1]; nz += q[2];
  }
  for (int w = 32; w; w >>= 1) { np += __shfl_xor(np, w); nn += __shfl_xor(nn, w); nz += __shfl_xor(nz, w); }
  if (lane != 0) return;
  S.npos = np; S.nneg = nn; S.nbad = nz;
  if (S.npos == D.nv && S.nbad == 0) {
    S.refactor = 0;
    if (S.delta_w > 0) S.delta_w_last = S.delta_w;
    if (S.mode == 0) S.ic_hot = S.delta_w > 0;
    return;
  }
  if (S.delta_w == 0.0) S.delta_w = S.delta_w_last == 0.0 ? o.delta_w_first : fmax(o.delta_w_min, o.kw_dec * S.delta_w_last);
  else S.delta_w *= S.delta_w_last == 0.0 ? o.kw_inc_first : o.kw_inc;
  if (S.delta_w > o.delta_w_max) { S.status = 4; return; }
  atomicAdd(&D.cnt[1], 1);
}

// ------------------------------------------------------------------------------------------------ direction
// fraction to the boundary (15): largest a in (0, 1] with w + a dw >= (1 - tau) w
__device__ inline double ftb(double w, double dw, double tau, double a) { return dw < 0 ? fmin(a, -tau * w / dw) : a; }

// step of the regular iteration from a solution of (13): dz (12), step lengths (15), slope of the barrier objective
__device__ inline bool newton_step(const IpmDev& D, int bi, const double* sol, double* dv, double* dlam, double* dzL, double* dzU,
                                   double mu, double tau, double* sh, double* amax_o, double* az_o, double* dphi_o, double* bad_o) {
  const size_t o = size_t(bi) * D.nv;
  const int i0 = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;    // this workgroup's slice (vec_combine)
  double amax = 1.0, az = 1.0, dphi = 0.0, bad = 0.0;
  #pragma unroll 4
  for (int i = i0; i < D.nv; i += stride) {
    const double l = D.vl[o + i], u = D.vu[o + i], vi = D.v[o + i], dsol = sol[D.pos[i]], zl = D.zL[o + i], zu = D.zU[o + i];
    const double gri = i < D.n ? D.grad[size_t(bi) * D.n + i] : 0.0;     // every load ahead of the branches (kept in flight by the unrolling)
    double d = 0.0, dl = 0.0, du = 0.0;
    if (l != u) {
      d = dsol;
      if (!(fabs(d) < 1e300)) bad = 1;
      double gphi = gri;
      if (l > -IPM_INF) {
        const double s = vi - l, z = zl;
        dl = mu / s - z - z / s * d;                       // (12)
        amax = ftb(s, d, tau, amax);                       // (15a)
        az = ftb(z, dl, tau, az);                          // (15b)
        gphi -= mu / s;
      }
      if (u < IPM_INF) {
        const double s = u - vi, z = zu;
        du = mu / s - z + z / s * d;
        amax = ftb(s, -d, tau, amax);
        az = ftb(z, du, tau, az);
        gphi += mu / s;
      }
      dphi += gphi * d;
    }
    dv[o + i] = d;
    dzL[o + i] = dl;
    dzU[o + i] = du;
  }
  #pragma unroll 4
  for (int r = i0; r < D.m; r += stride) dlam[size_t(bi) * D.m + r] = sol[D.pos[D.nv + r]];
  double vals[4] = {block_red(amax, 2, sh), block_red(az, 2, sh), block_red(dphi, 0, sh), block_red(bad, 1, sh)};
  const int kind[4] = {2, 2, 0, 1};
  if (!vec_combine(D, bi, vals, kind)) return false;       // only the last workgroup of the instance goes on
  *amax_o = vals[0]; *az_o = vals[1]; *dphi_o = vals[2]; *bad_o = vals[3];
  return true;
}
__device__ inline double alpha_min23(const IpmOpts& op, double theta, double theta_min, double dphi) {   // (23)
  double amin = op.gamma_theta;
  if (dphi < 0) {
    amin = fmin(amin, op.gamma_phi * theta / (-dphi));
    if (theta <= theta_min) amin = fmin(amin, op.delta * pow(theta, op.s_theta) / pow(-dphi, op.s_phi));
  }
  return op.gamma_alpha * amin;
}

__global__ __launch_bounds__(1024) void ipm_direction_kernel(IpmDev D) {
  __shared__ double sh[16];
  const int bi = blockIdx.y, t = threadIdx.x, i0 = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  IpmInst& S = D.inst[bi];
  const int status = S.status, mode = S.mode;
  const double mu = mode == 2 ? S.mu_r : S.mu, tau = S.tau;
  __syncthreads();              // thread 0 (of the last workgroup, vec_combine) rewrites S.mode below: everybody has read it
  if (status != 0) return;
  const size_t o = size_t(bi) * D.nv, om = size_t(bi) * D.m;
  const double* sol = D.rhs + size_t(bi) * D.Nt;
  if (mode == 3) {              // least-squares multipliers on leaving the restoration; lambda = 0 when they are large (section 3.6)
    double mx = 0.0;
    #pragma unroll 4
    for (int r = i0; r < D.m; r += stride) { const double w = fabs(sol[D.pos[D.nv + r]]); mx = (w < 1e300) ? fmax(mx, w) : 1e300; }   // NaN counts as too large
    double vals[1] = {block_red(mx, 1, sh)};
    const int kind[1] = {1};
    if (!vec_combine(D, bi, vals, kind)) return;
    const bool keep = vals[0] <= D.o.mult_reset;
    #pragma unroll 4
    for (int r = t; r < D.m; r += blockDim.x) D.lam[om + r] = keep ? sol[D.pos[D.nv + r]] : 0.0;
    if (t == 0) { S.mode = 0; S.accepted = 1; S.skip_update = S.skip_update == -1 ? 2 : (S.skip_update == -2 ? 3 : 1); S.ls = 0; S.armijo = 0; S.soc_on = 0; S.soc_req = 0; S.use_soc = 0; }
    return;
  }
  if (mode == 2) {              // restoration: step in (v, lambda) from the reduced system, p and n recovered from it
    const double rho = D.o.resto_rho, zeta = S.zeta;
    double amax = 1.0, az = 1.0, dphi = 0.0, bad = 0.0;
    #pragma unroll 4
    for (int i = i0; i < D.nv; i += stride) {
      const double l = D.vl[o + i], u = D.vu[o + i], vi = D.v[o + i];
      double d = 0.0, dl = 0.0, du = 0.0;
      if (l != u) {
        d = sol[D.pos[i]];
        if (!(fabs(d) < 1e300)) bad = 1;
        double gb = zeta * D.dr2[o + i] * (vi - D.vR[o + i]);
        if (l > -IPM_INF) {
          const double s = vi - l, z = D.zL[o + i];
          dl = mu / s - z - z / s * d;
          amax = ftb(s, d, tau, amax); az = ftb(z, dl, tau, az);
          gb -= mu / s;
        }
        if (u < IPM_INF) {
          const double s = u - vi, z = D.zU[o + i];
          du = mu / s - z + z / s * d;
          amax = ftb(s, -d, tau, amax); az = ftb(z, du, tau, az);
          gb += mu / s;
        }
        dphi += gb * d;
      }
      D.dv[o + i] = d; D.dzL[o + i] = dl; D.dzU[o + i] = du;
    }
    #pragma unroll 4
    for (int r = i0; r < D.m; r += stride) {
      const double pp = D.pp[om + r], nn = D.nn[om + r], zp = D.zp[om + r], zn = D.zn[om + r], lam = D.lam[om + r];
      const double sp = zp / pp, sn = zn / nn, dlam = sol[D.pos[D.nv + r]];
      if (!(fabs(dlam) < 1e300)) bad = 1;
      const double rp = rho - lam - mu / pp, rn = rho + lam - mu / nn;
      const double dp = (dlam - rp) / sp, dn = (-dlam - rn) / sn;
      const double dzp = mu / pp - zp - sp * dp, dzn = mu / nn - zn - sn * dn;
      amax = ftb(pp, dp, tau, amax); amax = ftb(nn, dn, tau, amax);
      az = ftb(zp, dzp, tau, az); az = ftb(zn, dzn, tau, az);
      dphi += (rho - mu / pp) * dp + (rho - mu / nn) * dn;
      D.dlam[om + r] = dlam; D.dpp[om + r] = dp; D.dnn[om + r] = dn; D.dzp[om + r] = dzp; D.dzn[om + r] = dzn;
    }
    {
      double vals[4] = {block_red(amax, 2, sh), block_red(az, 2, sh), block_red(dphi, 0, sh), block_red(bad, 1, sh)};
      const int kind[4] = {2, 2, 0, 1};
      if (!vec_combine(D, bi, vals, kind)) return;
      amax = vals[0]; az = vals[1]; dphi = vals[2]; bad = vals[3];
    }
    if (t != 0) return;
    if (bad != 0) { S.status = 5; return; }
    S.alpha_max = amax; S.alpha_z = az; S.alpha = amax; S.dphi = dphi;
    S.alpha_min = alpha_min23(D.o, S.th_r, S.thr_min, dphi);
    S.ls = 0; S.accepted = 0; S.armijo = 0; S.soc_on = 0; S.soc_req = 0; S.use_soc = 0;
    atomicAdd(&D.cnt[2], 1);
    return;
  }
  double amax, az, dphi, bad;
  if (!newton_step(D, bi, sol, D.dv, D.dlam, D.dzL, D.dzU, mu, tau, sh, &amax, &az, &dphi, &bad)) return;
  if (t != 0) return;
  if (bad != 0) { S.status = 5; return; }
  S.alpha_max = amax; S.alpha_z = az; S.alpha = amax; S.dphi = dphi;
  S.alpha_min = alpha_min23(D.o, S.theta, S.theta_min, dphi);
  S.ls = 0; S.accepted = 0; S.armijo = 0; S.soc_on = 0; S.soc_req = 0; S.use_soc = 0; S.soc_p = 0;
  atomicAdd(&D.cnt[2], 1);
}

// ---- second-order correction (paper section 2.4, A-5.5 .. A-5.9): the same matrix, c replaced by c_soc = alpha c_soc + c(trial)
__global__ void ipm_soc_rhs_kernel(IpmDev D) {
  const int bi = blockIdx.y;
  const IpmInst& S = D.inst[bi];
  if (S.status != 0 || !S.soc_req) return;
  const size_t o = size_t(bi) * D.nv, om = size_t(bi) * D.m;
  const int stride = gridDim.x * blockDim.x, t0 = blockIdx.x * blockDim.x + threadIdx.x;
  double* rhs = D.rhs + size_t(bi) * D.Nt;
  const bool first = S.soc_p == 0;
  const double mix = first ? S.alpha : S.alpha_soc, mu = S.mu;
  for (int r = t0; r < D.m; r += stride) {
    const double cs = mix * (first ? D.c[om + r] : D.csoc[om + r]) + D.ct[om + r];
    D.csoc[om + r] = cs;
    rhs[D.pos[D.nv + r]] = -cs;
  }
  for (int i = t0; i < D.nv; i += stride) {
    const double l = D.vl[o + i], u = D.vu[o + i];
    double r = 0.0;
    if (l != u) {
      r = D.glag[o + i];
      if (l > -IPM_INF) r -= mu / (D.v[o + i] - l);
      if (u < IPM_INF) r += mu / (u - D.v[o + i]);
    }
    rhs[D.pos[i]] = -r;
  }
}
__global__ __launch_bounds__(1024) void ipm_soc_direction_kernel(IpmDev D) {
  __shared__ double sh[16];
  const int bi = blockIdx.y;
  IpmInst& S = D.inst[bi];
  const int go = S.status == 0 && S.soc_req;
  const double mu = S.mu, tau = S.tau;
  __syncthreads();
  if (!go) return;
  double amax, az, dphi, bad;
  if (!newton_step(D, bi, D.rhs + size_t(bi) * D.Nt, D.dv2, D.dlam2, D.dzL2, D.dzU2, mu, tau, sh, &amax, &az, &dphi, &bad)) return;
  if (threadIdx.x != 0) return;
  S.soc_req = 0;
  if (bad != 0) { S.soc_on = 0; S.alpha = 0.5 * S.alpha; S.ls += 1; return; }   // no usable correction: back to the plain backtracking
  S.alpha_soc = amax; S.az_soc = az;
}

// ------------------------------------------------------------------------------------------------ line search
__global__ void ipm_trial_kernel(IpmDev D) {
  const int bi = blockIdx.y;
  const IpmInst& S = D.inst[bi];
  if (S.status != 0 || S.accepted) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= D.n) return;
  const size_t o = size_t(bi) * D.nv + i;
  D.xt[size_t(bi) * D.n + i] = S.soc_on ? D.v[o] + S.alpha_soc * D.dv2[o] : D.v[o] + S.alpha * D.dv[o];
}
// (several workgroups per instance when a few large instances run, vec_combine: the logarithms of 74 k unknowns kept one
// workgroup's issue slots busy for 100 us on the metric problem)
__global__ __launch_bounds__(1024) void ipm_accept_kernel(IpmDev D) {
  __shared__ double sh[16];
  const int bi = blockIdx.y, t = threadIdx.x, i0 = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  IpmInst& S = D.inst[bi];
  const int go = S.status == 0 && !S.accepted, mode = S.mode, soc = S.soc_on;
  const double a = soc ? S.alpha_soc : S.alpha;
  __syncthreads();
  if (!go) return;
  const size_t o = size_t(bi) * D.nv, om = size_t(bi) * D.m;
  const double* dvp = soc ? D.dv2 : D.dv;
  double th = 0.0, ln = 0.0, bad = 0.0, qd = 0.0, spn = 0.0, lnpn = 0.0;
  const bool resto = mode == 2;
  #pragma unroll 4
  for (int i = i0; i < D.nv; i += stride) {
    const double l = D.vl[o + i], u = D.vu[o + i], vi = D.v[o + i], di = dvp[o + i];   // loads ahead of the branch: the unrolled
    if (l == u) continue;                                                                 // iterations keep 16 of them in flight
    const double vt = vi + a * di;
    if (l > -IPM_INF) ln += log(vt - l);
    if (u < IPM_INF) ln += log(u - vt);
    if (resto) { const double dd = vt - D.vR[o + i]; qd += D.dr2[o + i] * dd * dd; }
  }
  #pragma unroll 4
  for (int r = i0; r < D.m; r += stride) {
    const int s = D.row_slack[r];
    const double gr = D.gt[size_t(bi) * D.sg + r];
    const double glr = D.scal_on ? D.sc[om + r] * D.gl[r] : D.gl[r];
    double cr = s < 0 ? gr - glr : gr - (D.v[o + D.n + s] + a * dvp[o + D.n + s]);
    D.ct[om + r] = cr;
    if (resto) {
      const double pt = D.pp[om + r] + a * D.dpp[om + r], nt = D.nn[om + r] + a * D.dnn[om + r];
      cr += nt - pt;
      spn += pt + nt;
      lnpn += log(pt) + log(nt);
    }
    if (!(fabs(cr) < 1e300)) bad = 1;
    th += fabs(cr);
  }
  th = block_red(th, 0, sh); ln = block_red(ln, 0, sh); bad = block_red(bad, 1, sh);
  if (resto) { qd = block_red(qd, 0, sh); spn = block_red(spn, 0, sh); lnpn = block_red(lnpn, 0, sh); }
  {
    double vals[6] = {th, ln, bad, qd, spn, lnpn};
    const int kind[6] = {0, 0, 1, 0, 0, 0};
    if (!vec_combine(D, bi, vals, kind)) return;
    th = vals[0]; ln = vals[1]; bad = vals[2]; qd = vals[3]; spn = vals[4]; lnpn = vals[5];
  }
  const IpmOpts& op = D.o;
  // is the trial point dominated by a filter entry?  (every thread holds the reduced sums; the entries — hundreds on a long
  // Delta-III solve — are dealt to the threads instead of being walked by thread 0)
  const double phit_all = resto ? op.resto_rho * spn + 0.5 * S.zeta * qd - S.mu_r * (ln + lnpn) : D.objt[bi] - S.mu * ln;
  double dom = 0.0;
  {
    const double* F = (resto ? D.rfilt : D.filt) + size_t(bi) * 2 * IPM_FMAX;
    const int nf = resto ? S.nrfilt : S.nfilt;
    for (int k = t; k < nf; k += blockDim.x)
      if (th >= F[2 * k] && phit_all >= F[2 * k + 1]) dom = 1.0;
  }
  const bool dominated = block_red(dom, 1, sh) != 0.0;
  if (t != 0) return;
  if (resto) {                  // the restoration problem's own filter line search
    const double phit = phit_all;
    const double slack = 10.0 * 2.220446049250313e-16 * fabs(S.phi_r);
    bool ok = false;
    if (bad == 0 && fabs(phit) < 1e300 && th <= S.thr_max) {
      if (!dominated) {
        const bool sw = S.dphi < 0 && a * pow(-S.dphi, op.s_phi) > op.delta * pow(S.th_r, op.s_theta);
        if (S.th_r <= S.thr_min && sw) {
          ok = phit - S.phi_r - op.eta_phi * a * S.dphi <= slack;
          if (ok) S.armijo = 1;
        } else {
          ok = th <= (1.0 - op.gamma_theta) * S.th_r || phit - (S.phi_r - op.gamma_phi * S.th_r) <= slack;
        }
      }
    }
    if (ok) { S.accepted = 1; return; }
    S.alpha = 0.5 * a;
    S.ls += 1;
    if (S.alpha < S.alpha_min || S.ls > op.max_ls) { S.status = 3; return; }     // the restoration failed
    atomicAdd(&D.cnt[2], 1);
    return;
  }
  const double ft = D.objt[bi];
  if (!(fabs(ft) < 1e300) || !(fabs(ln) < 1e300)) bad = 1;
  const double phit = phit_all;
  const double slack = 10.0 * 2.220446049250313e-16 * fabs(S.phi);     // Ipopt's rounding allowance in the phi comparisons
  const double a_test = S.alpha;       // the switching / Armijo tests of a corrected step use the uncorrected step length (A-5.7)
  bool ok = false;
  if (bad == 0 && th <= S.theta_max) {
    if (!dominated) {
      const bool sw = S.dphi < 0 && a_test * pow(-S.dphi, op.s_phi) > op.delta * pow(S.theta, op.s_theta);   // (19)
      if (S.theta <= S.theta_min && sw) {
        ok = phit - S.phi - op.eta_phi * a_test * S.dphi <= slack;                                            // (20)
        if (ok) S.armijo = 1;
      } else {
        ok = th <= (1.0 - op.gamma_theta) * S.theta || phit - (S.phi - op.gamma_phi * S.theta) <= slack;      // (18)
      }
    }
  }
  if (ok) {
    S.accepted = 1;
    if (soc) { S.use_soc = 1; S.n_soc += 1; }
    return;
  }
  const bool th_ok = fabs(th) < 1e300 && bad == 0;
  if (soc) {
    if (th_ok && S.soc_p + 1 < op.max_soc && th <= op.kappa_soc * S.th_old_soc) {      // A-5.9: next correction
      S.soc_p += 1; S.th_old_soc = th; S.soc_req = 1;
      atomicAdd(&D.cnt[3], 1);
      return;
    }
    S.soc_on = 0;                                                                       // give up: plain backtracking
  } else if (S.ls == 0 && op.max_soc > 0 && th_ok && th >= S.theta) {                   // A-5.5
    S.soc_on = 1; S.soc_p = 0; S.th_old_soc = S.theta; S.soc_req = 1;
    atomicAdd(&D.cnt[3], 1);
    return;
  }
  S.alpha = 0.5 * S.alpha;
  S.ls += 1;
  if (S.alpha < S.alpha_min || S.ls > op.max_ls) {
    if (S.err0 <= op.acceptable_tol) S.status = 6;             // nothing left to gain: Ipopt reports the acceptable level here too
    else if (op.resto && S.theta > op.tol) S.enter_resto = 1;  // Ipopt switches to its restoration phase here
    else if (op.resto && S.n_recalc < 3) S.enter_resto = 2;    // feasible but the multipliers are off: recompute them (recalc_y)
    else S.status = 3;
    return;
  }
  atomicAdd(&D.cnt[2], 1);
}

// ------------------------------------------------------------------------------------------------ step
__global__ __launch_bounds__(1024) void ipm_update_kernel(IpmDev D) {
  const int bi = blockIdx.y, t = threadIdx.x, i0 = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  IpmInst& S = D.inst[bi];
  // every thread reads the instance's verdicts BEFORE anybody changes any of them (a late wave must not see enter_resto already
  // cleared, or mode already switched, and skip its slice): the record is only written by thread 0 of the LAST workgroup of the
  // instance to get here (vec_combine with nothing to combine: the arrival ticket alone)
  double none[1] = {0.0};
  const int none_kind[1] = {0};
  const int s_status = S.status, s_enter = S.enter_resto, s_accepted = S.accepted, s_mode = S.mode, s_skip = S.skip_update, s_soc = S.use_soc;
  const double s_alpha = s_soc ? S.alpha_soc : S.alpha, s_alpha_z = s_soc ? S.az_soc : S.alpha_z, s_mu = S.mu, s_mu_r = S.mu_r, s_cinf = S.cinf;
  __syncthreads();
  if (s_status != 0) return;
  const size_t o = size_t(bi) * D.nv, om = size_t(bi) * D.m;
  const double ks = D.o.kappa_sigma;
  if (s_skip) {           // this pass only replaced lambda (least-squares multipliers after the restoration, or recalc_y)
    if (!vec_combine(D, bi, none, none_kind)) return;
    if (t == 0) { if (s_skip == 1) S.n_resto += 1; else if (s_skip == 2) S.n_recalc += 1; S.skip_update = 0; }
    return;
  }
  if (s_enter == 2) {     // recalc_y: the next pass computes least-squares multipliers at this point, nothing else
    if (!vec_combine(D, bi, none, none_kind)) return;
    if (t == 0) { S.mode = 3; S.enter_resto = 0; S.skip_update = -1; }
    return;
  }
  if (s_enter) {          // the line search gave up at an infeasible point: start the restoration phase from it
    const double rho = D.o.resto_rho, mu_r = fmax(s_mu, s_cinf);
    #pragma unroll 4
    for (int i = i0; i < D.nv; i += stride) {
      const double vi = D.v[o + i], sc = fmax(1.0, fabs(vi));
      D.vR[o + i] = vi;
      D.dr2[o + i] = 1.0 / (sc * sc);
      D.zL[o + i] = fmin(rho, D.zL[o + i]);
      D.zU[o + i] = fmin(rho, D.zU[o + i]);
    }
    #pragma unroll 4
    for (int r = i0; r < D.m; r += stride) {        // (33), (34): the p, n that minimise the restoration's barrier objective at v_R
      const double c = D.c[om + r], h2 = (mu_r - rho * c) / (2.0 * rho);
      const double nn = h2 + sqrt(h2 * h2 + mu_r * c / (2.0 * rho)), pp = c + nn;
      D.nn[om + r] = nn; D.pp[om + r] = pp;
      D.zp[om + r] = mu_r / pp; D.zn[om + r] = mu_r / nn;
      D.lam[om + r] = 0.0;
    }
    if (!vec_combine(D, bi, none, none_kind)) return;
    if (t == 0) {
      if (S.nfilt < IPM_FMAX) {
        double* F = D.filt + size_t(bi) * 2 * IPM_FMAX;
        F[2 * S.nfilt] = (1.0 - D.o.gamma_theta) * S.theta;
        F[2 * S.nfilt + 1] = S.phi - D.o.gamma_phi * S.theta;
        S.nfilt += 1;
      }
      S.th0 = S.theta; S.mu_r = mu_r; S.zeta = sqrt(mu_r); S.mode = 2; S.resto_it = 0; S.enter_resto = 0; S.nrfilt = 0;
    }
    return;
  }
  if (!s_accepted) return;
  if (s_mode == 2) {
    const double a = s_alpha, az = s_alpha_z, mu = s_mu_r;
    #pragma unroll 4
    for (int i = i0; i < D.nv; i += stride) {
      const double l = D.vl[o + i], u = D.vu[o + i];
      if (l == u) continue;
      const double vi = D.v[o + i] + a * D.dv[o + i];
      D.v[o + i] = vi;
      if (l > -IPM_INF) D.zL[o + i] = reset16(D.zL[o + i] + az * D.dzL[o + i], vi - l, mu, ks);
      if (u < IPM_INF) D.zU[o + i] = reset16(D.zU[o + i] + az * D.dzU[o + i], u - vi, mu, ks);
    }
    #pragma unroll 4
    for (int r = i0; r < D.m; r += stride) {
      const double pp = D.pp[om + r] + a * D.dpp[om + r], nn = D.nn[om + r] + a * D.dnn[om + r];
      D.pp[om + r] = pp; D.nn[om + r] = nn;
      D.zp[om + r] = reset16(D.zp[om + r] + az * D.dzp[om + r], pp, mu, ks);
      D.zn[om + r] = reset16(D.zn[om + r] + az * D.dzn[om + r], nn, mu, ks);
      D.lam[om + r] += a * D.dlam[om + r];
    }
    if (!vec_combine(D, bi, none, none_kind)) return;
    if (t == 0) {
      if (!S.armijo && S.nrfilt < IPM_FMAX) {
        double* F = D.rfilt + size_t(bi) * 2 * IPM_FMAX;
        F[2 * S.nrfilt] = (1.0 - D.o.gamma_theta) * S.th_r;
        F[2 * S.nrfilt + 1] = S.phi_r - D.o.gamma_phi * S.th_r;
        S.nrfilt += 1;
      }
      if (D.trace && S.iter < D.trace_cap) {
        double* R = D.trace + (size_t(bi) * D.trace_cap + S.iter) * IPM_TRACE;
        R[0] = S.f; R[1] = S.theta; R[2] = S.mu_r; R[3] = a; R[4] = az; R[5] = 0.0; R[6] = S.err0; R[7] = -1.0;
      }
      S.resto_it += 1;
      S.iter += 1;
    }
    return;
  }
  const double a = s_alpha, az = s_alpha_z, mu = s_mu;
  const double *dv = s_soc ? D.dv2 : D.dv, *dlam = s_soc ? D.dlam2 : D.dlam, *dzL = s_soc ? D.dzL2 : D.dzL, *dzU = s_soc ? D.dzU2 : D.dzU;
  #pragma unroll 4
  for (int i = i0; i < D.nv; i += stride) {
    const double l = D.vl[o + i], u = D.vu[o + i], v0 = D.v[o + i], di = dv[o + i];
    const double zl = D.zL[o + i], zu = D.zU[o + i], dl = dzL[o + i], du = dzU[o + i];        // loads ahead of the branch
    if (l == u) continue;
    const double vi = v0 + a * di;
    D.v[o + i] = vi;
    if (l > -IPM_INF) D.zL[o + i] = reset16(zl + az * dl, vi - l, mu, ks);   // (16)
    if (u < IPM_INF) D.zU[o + i] = reset16(zu + az * du, u - vi, mu, ks);
  }
  #pragma unroll 4
  for (int r = i0; r < D.m; r += stride) D.lam[om + r] += a * dlam[om + r];
  if (!vec_combine(D, bi, none, none_kind)) return;
  if (t == 0) {
    if (!S.armijo && S.nfilt < IPM_FMAX) {       // (22)
      double* F = D.filt + size_t(bi) * 2 * IPM_FMAX;
      F[2 * S.nfilt] = (1.0 - D.o.gamma_theta) * S.theta;
      F[2 * S.nfilt + 1] = S.phi - D.o.gamma_phi * S.theta;
      S.nfilt += 1;
    }
    if (D.trace && S.iter < D.trace_cap) {
      double* R = D.trace + (size_t(bi) * D.trace_cap + S.iter) * IPM_TRACE;
      R[0] = S.f; R[1] = S.theta; R[2] = S.mu; R[3] = a; R[4] = az; R[5] = S.delta_w; R[6] = S.err0; R[7] = double(S.ls);
    }
    S.iter += 1;
  }
}


// ------------------------------------------------------------------------------------------------ launchers
// threads per instance of the one-workgroup-per-instance vector kernels: a few large instances (the metric problem: n = 41 k)
// get 16 waves each, a sweep of many small ones 4
static unsigned vec_threads(const IpmDev& D) { return D.B <= 32 && D.nv >= 4096 ? 1024u : 256u; }
// workgroups per instance of the vector kernels that can split an instance (ipm_accept_kernel)
static unsigned vec_blocks(const IpmDev& D) {
  return D.B <= 32 && D.nv >= 4096 ? unsigned(std::min(IPM_VEC_BLOCKS, (std::max(D.nv, D.m) + 1023) / 1024)) : 1u;
}
void ipm_launch_init(const IpmDev& D, const double* d_x0, hipStream_t st) {
  hipLaunchKernelGGL(ipm_init_kernel, dim3(unsigned(D.B)), dim3(256), 0, st, D, d_x0);
}
void ipm_launch_init_slack(const IpmDev& D, hipStream_t st) {
  hipLaunchKernelGGL(ipm_init_slack_kernel, dim3(unsigned(D.B)), dim3(256), 0, st, D);
}
void ipm_launch_pack_x(const IpmDev& D, hipStream_t st) {
  hipLaunchKernelGGL(ipm_pack_x_kernel, dim3(unsigned((D.n + 255) / 256), unsigned(D.B)), dim3(256), 0, st, D);
}
void ipm_launch_residual(const IpmDev& D, hipStream_t st) {
  const int tb = (D.nv + 255) / 256;
  hipLaunchKernelGGL(ipm_jt_lambda_kernel, dim3(unsigned(tb + D.n_long), unsigned(D.B)), dim3(256), 0, st, D, tb);
  hipLaunchKernelGGL(ipm_residual_kernel, dim3(vec_blocks(D), unsigned(D.B)), dim3(vec_threads(D)), 0, st, D);
}
void ipm_launch_jt_lambda_into(const IpmDev& D, double* out, hipStream_t st) {
  IpmDev D2 = D;
  D2.glag = out;
  const int tb = (D.nv + 255) / 256;
  hipLaunchKernelGGL(ipm_jt_lambda_kernel, dim3(unsigned(tb + D.n_long), unsigned(D.B)), dim3(256), 0, st, D2, tb);
}
void ipm_launch_assemble(const IpmDev& D, int nnz_max, hipStream_t st) {
  if (D.as_nchunk > 0) {
    const int chunks = kkt_level1_fused(D) ? D.as_nlive : D.as_nchunk;
    hipLaunchKernelGGL(ipm_fill_kernel, dim3(unsigned(std::max(1, std::min(chunks, 65535))), unsigned(D.B)), dim3(256), 0, st, D);
    return;
  }
  const int assemble_blocks = std::max(1, std::min(D.B <= 32 ? 2048 : 64, (nnz_max + 255) / 256));   // a few large instances: the whole chip
  const int zero_blocks = int(std::max<long long>(1, std::min<long long>(256, D.kstride / 2 / 256 + 1)));
  hipLaunchKernelGGL(ipm_zero_kernel, dim3(unsigned(zero_blocks), unsigned(D.B)), dim3(256), 0, st, D);
  hipLaunchKernelGGL(ipm_assemble_kernel, dim3(unsigned(assemble_blocks), unsigned(D.B)), dim3(256), 0, st, D);
}
void ipm_launch_inertia(const IpmDev& D, hipStream_t st) {
  hipLaunchKernelGGL(ipm_inertia_kernel, dim3(unsigned((D.B + 3) / 4)), dim3(256), 0, st, D);
}
void ipm_launch_direction(const IpmDev& D, hipStream_t st) {
  hipLaunchKernelGGL(ipm_direction_kernel, dim3(vec_blocks(D), unsigned(D.B)), dim3(vec_threads(D)), 0, st, D);
}
void ipm_launch_trial(const IpmDev& D, hipStream_t st) {
  hipLaunchKernelGGL(ipm_trial_kernel, dim3(unsigned((D.n + 255) / 256), unsigned(D.B)), dim3(256), 0, st, D);
}
void ipm_launch_accept(const IpmDev& D, hipStream_t st) {
  hipLaunchKernelGGL(ipm_accept_kernel, dim3(vec_blocks(D), unsigned(D.B)), dim3(vec_threads(D)), 0, st, D);
}
void ipm_launch_update(const IpmDev& D, hipStream_t st) {
  hipLaunchKernelGGL(ipm_update_kernel, dim3(vec_blocks(D), unsigned(D.B)), dim3(vec_threads(D)), 0, st, D);
}
void ipm_launch_soc_rhs(const IpmDev& D, hipStream_t st) {
  const int blocks = std::max(1, std::min(64, (std::max(D.nv, D.m) + 255) / 256));
  hipLaunchKernelGGL(ipm_soc_rhs_kernel, dim3(unsigned(blocks), unsigned(D.B)), dim3(256), 0, st, D);
}
void ipm_launch_soc_direction(const IpmDev& D, hipStream_t st) {
  hipLaunchKernelGGL(ipm_soc_direction_kernel, dim3(vec_blocks(D), unsigned(D.B)), dim3(vec_threads(D)), 0, st, D);
}
// ------------------------------------------------------------------------------------------------ NLP scaling
// Ipopt's GradientScaling (nlp_scaling_method = gradient-based, its default; option nlp_scaling here): at the caller's starting
// point, sf = min(1, gmax / |grad f|_inf) and sc_i = min(1, gmax / |grad c_i|_inf) over the free variables (floor scal_min); the
// solver then works on sf f and sc o c — values scaled in place right after every evaluation — and hands back lambda o sc / sf.
__global__ void ipm_scal_max_kernel(IpmDev D) {      // row maxima into sc, gradient maximum into sf (as bit patterns of non-negative doubles)
  const int bi = blockIdx.y;
  unsigned long long* rmax = reinterpret_cast<unsigned long long*>(D.sc + size_t(bi) * D.m);
  const double* jac = D.jac + size_t(bi) * D.sv;
  const int i0 = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  for (int k = i0; k < D.nnz_jac; k += stride)
    if (D.jac_dst[k] >= 0) atomicMax(&rmax[D.jac_row[k]], (unsigned long long)__double_as_longlong(fabs(jac[k])));
  double gm = 0.0;
  for (int i = i0; i < D.n; i += stride)
    if (D.vl[size_t(bi) * D.nv + i] != D.vu[size_t(bi) * D.nv + i]) gm = fmax(gm, fabs(D.grad[size_t(bi) * D.n + i]));
  if (gm > 0.0) atomicMax(reinterpret_cast<unsigned long long*>(D.sf + bi), (unsigned long long)__double_as_longlong(gm));
}
__global__ void ipm_scal_finish_kernel(IpmDev D) {
  const int bi = blockIdx.y;
  const double gmax = D.o.scal_gmax, vmin = D.o.scal_min;
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < D.m; r += gridDim.x * blockDim.x) {
    const double v = D.sc[size_t(bi) * D.m + r];
    D.sc[size_t(bi) * D.m + r] = v > gmax ? fmax(gmax / v, vmin) : 1.0;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const double v = D.sf[bi];
    D.sf[bi] = v > gmax ? fmax(gmax / v, vmin) : 1.0;
  }
}
__global__ void ipm_scal_apply_kernel(IpmDev D, double* g, double* jac, int jac0, int jac1, double* obj, double* grad) {
  const int bi = blockIdx.y;
  if (D.inst[bi].status != 0) return;
  const double* sc = D.sc + size_t(bi) * D.m;
  const double sf = D.sf[bi];
  const int i0 = blockIdx.x * blockDim.x + threadIdx.x, stride = gridDim.x * blockDim.x;
  if (g) for (int r = i0; r < D.m; r += stride) g[size_t(bi) * D.sg + r] *= sc[r];
  if (jac) for (int k = jac0 + i0; k < jac1; k += stride) jac[size_t(bi) * D.sv + k] *= sc[D.jac_row[k]];
  if (grad) for (int i = i0; i < D.n; i += stride) grad[size_t(bi) * D.n + i] *= sf;
  if (obj && i0 == 0) obj[bi] *= sf;
}
__global__ void ipm_scal_lambda_kernel(IpmDev D, double* out) {   // lambda o sc / sf: the multipliers of the unscaled rows over the objective's factor
  const int bi = blockIdx.y;
  const double sf = D.sf[bi];
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < D.m; r += gridDim.x * blockDim.x)
    out[size_t(bi) * D.m + r] = D.lam[size_t(bi) * D.m + r] * D.sc[size_t(bi) * D.m + r] / sf;
}
__global__ void ipm_scal_hess_kernel(IpmDev D) {
  const int bi = blockIdx.y;
  if (D.inst[bi].status != 0) return;
  const double sf = D.sf[bi];
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < D.nnz_h; k += gridDim.x * blockDim.x) D.hess[size_t(bi) * D.nnz_h + k] *= sf;
}
static dim3 scal_grid(const IpmDev& D, int n) { return dim3(unsigned(std::max(1, std::min(D.B <= 32 ? 256 : 16, (n + 255) / 256))), unsigned(D.B)); }
void ipm_launch_scaling_factors(const IpmDev& D, hipStream_t st) {
  (void)hipMemsetAsync(D.sc, 0, size_t(D.B) * D.m * sizeof(double), st);
  (void)hipMemsetAsync(D.sf, 0, size_t(D.B) * sizeof(double), st);
  hipLaunchKernelGGL(ipm_scal_max_kernel, scal_grid(D, std::max(D.nnz_jac, D.n)), dim3(256), 0, st, D);
  hipLaunchKernelGGL(ipm_scal_finish_kernel, scal_grid(D, D.m), dim3(256), 0, st, D);
}
void ipm_launch_scale(const IpmDev& D, double* g, double* jac, int jac0, int jac1, double* obj, double* grad, hipStream_t st) {
  const int n = std::max(std::max(g ? D.m : 0, jac ? jac1 - jac0 : 0), std::max(grad ? D.n : 0, 1));
  hipLaunchKernelGGL(ipm_scal_apply_kernel, scal_grid(D, n), dim3(256), 0, st, D, g, jac, jac0, jac1, obj, grad);
}
void ipm_launch_scale_lambda(const IpmDev& D, hipStream_t st) {
  hipLaunchKernelGGL(ipm_scal_lambda_kernel, scal_grid(D, D.m), dim3(256), 0, st, D, D.lam_h);
}
void ipm_launch_scale_hessian(const IpmDev& D, hipStream_t st) {
  hipLaunchKernelGGL(ipm_scal_hess_kernel, scal_grid(D, D.nnz_h), dim3(256), 0, st, D);
}
void ipm_launch_unscale_lambda(const IpmDev& D, double* out, hipStream_t st) {
  hipLaunchKernelGGL(ipm_scal_lambda_kernel, scal_grid(D, D.m), dim3(256), 0, st, D, out);
}

size_t kkt_factor_lds_bytes(const IpmPlan& p) {
  if (p.nd) return p.max_factor_lds;
  return (size_t(p.b + 24) * IPM_W + size_t(IPM_W) * (IPM_W + 1) + size_t(IPM_W) * IPM_W + IPM_W + 2 * size_t(p.nb) * IPM_W +
          size_t(p.nb) * (p.nb + 1) / 2) * sizeof(double);
}
hipError_t kkt_factor_prepare(int tiles_per_wave, size_t lds_bytes) {
  if (lds_bytes <= 48 * 1024) return hipSuccess;
  if (tiles_per_wave == 38) return hipFuncSetAttribute(reinterpret_cast<const void*>(kkt_factor_kernel<3, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes));
  if (tiles_per_wave == 28) return hipFuncSetAttribute(reinterpret_cast<const void*>(kkt_factor_kernel<2, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes));
  return hipFuncSetAttribute(tiles_per_wave == 4 ? reinterpret_cast<const void*>(kkt_factor_kernel<4>)
                             : tiles_per_wave == 6 ? reinterpret_cast<const void*>(kkt_factor_kernel<6>)
                                                   : reinterpret_cast<const void*>(kkt_factor_kernel<IPM_MT>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_bytes));
}
// ---- nested dissection glue: sums in a fixed (interval) order, so the factorisation stays deterministic ----------------------
// K[dst[i]] += sum_j K[src[j]], j in [ptr[i], ptr[i+1]): the level-1 Schur complements into the separator system
// Destinations are sorted by the length of their source lists (build_ipm_plan_nd): the first n_long of them (>= 32 sources: the
// global border's corner entries collect one contribution from EVERY interval, 256 on the metric problem) take a wave each —
// lanes stride the list, partial sums combined in a fixed butterfly order — the rest a thread each.
__global__ void kkt_gather_add_kernel(double* Kall, long long kstride, const int* __restrict__ ptr, const int* __restrict__ src,
                                      const int* __restrict__ dst, int n, const IpmInst* inst, int need_refactor, int n_long, int kmod) {
  const int bi = blockIdx.y, bk = bi % kmod;
  if (inst[bk].status != 0 || (need_refactor && !inst[bk].refactor)) return;
  double* K = Kall + size_t(bi) * kstride;
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
  for (int i = wave; i < n_long; i += n_waves) {
    double acc = 0.0;
    for (int j = ptr[i] + lane; j < ptr[i + 1]; j += 64) acc += K[src[j]];
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) K[dst[i]] += acc;
  }
  for (int i = n_long + blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    double acc = K[dst[i]];
    for (int j = ptr[i]; j < ptr[i + 1]; ++j) acc += K[src[j]];
    K[dst[i]] = acc;
  }
}
// v[dst[i]] += v[src[ptr[i]]] + v[src[ptr[i] + 1]] + ... in list order, a WAVE per destination: 64 sources are fetched at a time, then
// every lane adds them up in order through shuffles (lane 0 stores).  The right-hand-side gather of the nested dissection: the
// global border's rows collect one term from every interval — 256 on the metric problem, 88 us as 256 dependent loads of one
// thread, a few microseconds this way — with the sums' order, and so their bits, unchanged.
__global__ __launch_bounds__(256) void kkt_gather_seq_kernel(double* vall, long long vstride, const int* __restrict__ ptr, const int* __restrict__ src,
                                                             const int* __restrict__ dst, int n, const IpmInst* inst, int check_status, int kmod) {
  const int bi = blockIdx.y, bk = bi % kmod;
  if (check_status && (inst[bk].status != 0 || (check_status == 2 && !inst[bk].soc_req))) return;
  double* v = vall + size_t(bi) * vstride;
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
  for (int i = wave; i < n; i += n_waves) {
    const int j0 = ptr[i], j1 = ptr[i + 1];
    double acc = v[dst[i]];
    for (int j = j0; j < j1; j += 64) {
      const double mine = j + lane < j1 ? v[src[j + lane]] : 0.0;
      const int m = min(64, j1 - j);
      for (int k = 0; k < m; ++k) acc += __shfl(mine, k, 64);
    }
    if (lane == 0) v[dst[i]] = acc;
  }
}
// mode 0: v[pos[i]] = 0;  mode 1: v[dst[i]] = v[src[i]]
__global__ void kkt_vec_kernel(double* vall, long long vstride, const int* __restrict__ dst, const int* __restrict__ src, int n, int mode,
                               const IpmInst* inst, int check_status, int kmod) {
  const int bi = blockIdx.y, bk = bi % kmod;
  if (check_status && (inst[bk].status != 0 || (check_status == 2 && !inst[bk].soc_req) || (check_status == 3 && !inst[bk].refactor))) return;
  double* v = vall + size_t(bi) * vstride;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) v[dst[i]] = mode ? v[src[i]] : 0.0;
}

// level 1 assembled and forward-substituted inside kkt_factor_dense_kernel
int kkt_level1_fused(const IpmDev& D) { return (D.n_l1 > 0 && D.l1_dense_lds && D.df_on && D.df_map) ? 1 : 0; }
static void launch_factor_subs(const IpmDev& D, int sub0, int n_here, int partial, int tiles_per_wave, size_t lds_bytes, hipStream_t st,
                               size_t dense_lds = 0) {
  const dim3 grid(unsigned(D.B) * unsigned(n_here));
  if (dense_lds) {   // every sub-problem of this level fits the register tiles
    // a last level: its corner there too (D.last_dense_corner; the panels of the corner's block columns by substitution, see the
    // kernel), or by kkt_factor_kernel's unblocked elimination (partial = 2: 143 us for the metric problem's 73 x 73)
    const bool corner = !partial && D.last_dense_corner;
    const int dense_partial = corner ? 0 : 1;
#define IPM_LAUNCH_DENSE(EARLY_, CORNER_)                                                                                          \
    hipLaunchKernelGGL((kkt_factor_dense_kernel<EARLY_, CORNER_>), grid, dim3(512), dense_lds, st, D.K, D.kstride, D.subs, sub0, n_here, D.n_sub, \
                       D.inst, D.piv, D, 0, 0, dense_partial)
    if (dense_early(dense_lds)) { if (corner) IPM_LAUNCH_DENSE(true, true); else IPM_LAUNCH_DENSE(true, false); }
    else { if (corner) IPM_LAUNCH_DENSE(false, true); else IPM_LAUNCH_DENSE(false, false); }
#undef IPM_LAUNCH_DENSE
    if (partial || corner) return;
    partial = 2;       // the corner alone, below
  }
  if (tiles_per_wave == 28)
    hipLaunchKernelGGL((kkt_factor_kernel<2, 8>), grid, dim3(512), lds_bytes, st, D.K, D.kstride, D.subs, sub0, n_here, D.n_sub, D.inst, D.piv, partial);
  else if (tiles_per_wave == 38)
    hipLaunchKernelGGL((kkt_factor_kernel<3, 8>), grid, dim3(512), lds_bytes, st, D.K, D.kstride, D.subs, sub0, n_here, D.n_sub, D.inst, D.piv, partial);
  else if (tiles_per_wave == 4)
    hipLaunchKernelGGL(kkt_factor_kernel<4>, grid, dim3(256), lds_bytes, st, D.K, D.kstride, D.subs, sub0, n_here, D.n_sub, D.inst, D.piv, partial);
  else if (tiles_per_wave == 6)
    hipLaunchKernelGGL(kkt_factor_kernel<6>, grid, dim3(256), lds_bytes, st, D.K, D.kstride, D.subs, sub0, n_here, D.n_sub, D.inst, D.piv, partial);
  else
    hipLaunchKernelGGL(kkt_factor_kernel<IPM_MT>, grid, dim3(256), lds_bytes, st, D.K, D.kstride, D.subs, sub0, n_here, D.n_sub, D.inst, D.piv,
                       partial);
}
static void launch_solve_subs(const IpmDev& D, int sub0, int n_here, int phase, int check_status, hipStream_t st) {
  const dim3 grid(unsigned(D.B) * unsigned(D.rhs_mult > 1 ? D.rhs_mult : 1) * unsigned(n_here));
  if (size_t(D.max_sub_nt) * sizeof(double) <= 48 * 1024 && grid.x <= 512)        // few workgroups: two steps' factor entries in flight
    hipLaunchKernelGGL((kkt_solve_kernel<true, 2>), grid, dim3(256), size_t(D.max_sub_nt) * sizeof(double), st, D.K, D.kstride, D.subs, sub0, n_here,
                       D.inst, D.rhs, (long long)D.Nt, check_status, phase, D.B);
  else if (size_t(D.max_sub_nt) * sizeof(double) <= 48 * 1024)
    hipLaunchKernelGGL(kkt_solve_kernel<true>, grid, dim3(256), size_t(D.max_sub_nt) * sizeof(double), st, D.K, D.kstride, D.subs, sub0, n_here,
                       D.inst, D.rhs, (long long)D.Nt, check_status, phase, D.B);
  else
    hipLaunchKernelGGL(kkt_solve_kernel<false>, grid, dim3(256), 0, st, D.K, D.kstride, D.subs, sub0, n_here, D.inst, D.rhs, (long long)D.Nt,
                       check_status, phase, D.B);
}
void kkt_launch_factor(const IpmDev& D, int tiles_per_wave, size_t lds_bytes, hipStream_t st) {
  if (D.n_l1 == 0) {
    launch_factor_subs(D, 0, 1, 0, tiles_per_wave, lds_bytes, st);
    return;
  }
  auto corners = [&](const int* ptr, const int* src, const int* dst, int n, int n_long) {
    if (!n) return;
    const unsigned blocks = unsigned(std::max(1, std::min(1024, (n + 255) / 256)));
    hipLaunchKernelGGL(kkt_gather_add_kernel, dim3(blocks, unsigned(D.B)), dim3(256), 0, st, D.K, D.kstride, ptr, src, dst, n, D.inst, 1, n_long, D.B);
  };
  const int fused = kkt_level1_fused(D);
  if (fused && D.n_gap) {                                                              // the forward sweep runs inside the kernel: border work spaces start at zero
    const unsigned blocks = unsigned(std::max(1, std::min(256, (D.n_gap + 255) / 256)));
    hipLaunchKernelGGL(kkt_vec_kernel, dim3(blocks, unsigned(D.B)), dim3(256), 0, st, D.rhs, (long long)D.Nt, D.gap_pos, nullptr, D.n_gap, 0, D.inst, 3, D.B);
  }
  if (D.l1_dense_lds && dense_early(D.l1_dense_lds))                                   // every interval up to its corner
    hipLaunchKernelGGL((kkt_factor_dense_kernel<true, false>), dim3(unsigned(D.B) * unsigned(D.n_l1)), dim3(512), D.l1_dense_lds, st, D.K, D.kstride, D.subs, 0,
                       D.n_l1, D.n_sub, D.inst, D.piv, D, fused, fused, 1);
  else if (D.l1_dense_lds)
    hipLaunchKernelGGL((kkt_factor_dense_kernel<false, false>), dim3(unsigned(D.B) * unsigned(D.n_l1)), dim3(512), D.l1_dense_lds, st, D.K, D.kstride, D.subs, 0,
                       D.n_l1, D.n_sub, D.inst, D.piv, D, fused, fused, 1);
  else
    launch_factor_subs(D, 0, D.n_l1, 1, tiles_per_wave, lds_bytes, st);
  corners(D.cg_ptr, D.cg_src, D.cg_dst, D.n_cg, D.n_cg_long);
  if (D.n_l2) {
    launch_factor_subs(D, D.n_l1, D.n_l2, 1, tiles_per_wave, lds_bytes, st, D.l2_dense_lds);   // every group of separators up to its corner
    corners(D.cg2_ptr, D.cg2_src, D.cg2_dst, D.n_cg2, D.n_cg2_long);
  }
  launch_factor_subs(D, D.n_l1 + D.n_l2, 1, 0, tiles_per_wave, lds_bytes, st, D.last_dense_lds);   // last level: (group) separators + border
}
void kkt_launch_solve(const IpmDev& D, int check_status, hipStream_t st, int forward_done) {
  const unsigned VB = unsigned(D.B) * unsigned(D.rhs_mult > 1 ? D.rhs_mult : 1);   // right-hand sides in D.rhs (rhs_mult per instance)
  if (D.n_l1 == 0) {
    launch_solve_subs(D, 0, 1, 0, check_status, st);
    return;
  }
  auto vec = [&](const int* dst, const int* src, int n, int mode) {
    if (!n) return;
    const unsigned blocks = unsigned(std::max(1, std::min(256, (n + 255) / 256)));
    hipLaunchKernelGGL(kkt_vec_kernel, dim3(blocks, VB), dim3(256), 0, st, D.rhs, (long long)D.Nt, dst, src, n, mode, D.inst,
                       check_status, D.B);
  };
  auto gather = [&](const int* ptr, const int* src, const int* dst, int n) {
    if (!n) return;
    const unsigned blocks = unsigned(std::max(1, std::min(4096, (n + 3) / 4)));           // a wave per destination
    hipLaunchKernelGGL(kkt_gather_seq_kernel, dim3(blocks, VB), dim3(256), 0, st, D.rhs, (long long)D.Nt, ptr, src, dst, n, D.inst, check_status, D.B);
  };
  if (!(forward_done && kkt_level1_fused(D))) {                                        // (else kkt_factor_dense_kernel did both for this right-hand side)
    vec(D.gap_pos, nullptr, D.n_gap, 0);                                               // border work spaces start at zero
    launch_solve_subs(D, 0, D.n_l1, 1, check_status, st);                              // forward, every interval
  }
  gather(D.rg_ptr, D.rg_src, D.rg_dst, D.n_rg);
  if (D.n_l2) {
    launch_solve_subs(D, D.n_l1, D.n_l2, 1, check_status, st);                         // forward, every group
    gather(D.rg2_ptr, D.rg2_src, D.rg2_dst, D.n_rg2);
  }
  launch_solve_subs(D, D.n_l1 + D.n_l2, 1, 0, check_status, st);                       // last level
  if (D.n_l2) {
    vec(D.rs2_dst, D.rs2_src, D.n_rs2, 1);                                             // its solution into the groups' work spaces
    launch_solve_subs(D, D.n_l1, D.n_l2, 2, check_status, st);                         // backward, every group
  }
  vec(D.rs_dst, D.rs_src, D.n_rs, 1);                                                  // separator / border values into the intervals' work spaces
  launch_solve_subs(D, 0, D.n_l1, 2, check_status, st);                                // backward, every interval
}

}  // namespace rpm

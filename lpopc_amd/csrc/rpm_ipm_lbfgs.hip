// rpm_ipm_lbfgs.hip — hessian-approximation = limited-memory on the device solver (row f-2): what the reference configures by
// default (Core/LpNLPWrapper.hpp:71, handed to Ipopt at Core/LpNLPSolver.cpp:27-33).  Ipopt is absent from the reference tree;
// this restates its LimMemQuasiNewtonUpdater with Ipopt 3.12's defaults (oracle/ipm_oracle.py has the same in numpy): BFGS,
// limited_memory_max_history 6, one pair per accepted step
//     s = x+ - x,   y = grad_x L(x+, lambda+) - grad_x L(x, lambda+)          (x only: the slacks' Hessian block is zero),
// skipped when s'y <= sqrt(eps) |s| |y| (two skips in a row empty the memory), scaling sigma = s'y / s's in [1e-8, 1e8], and the
// compact representation  B = sigma I - Q M^-1 Q',  Q = [sigma S  Y],  M = [[sigma S'S, L], [L', -D]]  (Byrd, Nocedal, Schnabel 1994)
// in the place of the exact Hessian W.  The KKT matrix then is  K = K0 - E M^-1 E'  with K0 the matrix of a DIAGONAL Hessian
// sigma I (what is assembled and factored: no Hessian entries, a narrower band) and E = Q at the positions of x; every solve is
//     d = z0 + Z (M - E'Z)^-1 E' z0,      z0 = K0^-1 r,   Z = K0^-1 E     (Sherman-Morrison-Woodbury; Ipopt's LowRankAugSystemSolver)
// with Z from 2 x history substitutions per iteration with the factors in place.  B is positive definite, so K has the right
// inertia whenever K0 has.  Fixed layout of the small matrices: index a < H = pair a's sigma s column, H + a = its y column; unused
// indices are identity rows, so one code path serves every fill level of the memory and every instance of a batch.
#include "rpm_device_internal.hpp"
#include "rpm_ipm_device.hpp"

namespace rpm {

constexpr int LB_TH = 2 * IPM_LB_H;
// per-instance record (doubles): 0 sigma, 1 pairs held, 2 consecutive skips, 3 previous iterate valid, 4 updates, 5 skips,
// 8.. M (TH x TH), then C = M - E'Z (LU in place), then its pivots (TH), then a work vector (TH)
constexpr int LB_M = 8, LB_C = LB_M + LB_TH * LB_TH, LB_PIV = LB_C + LB_TH * LB_TH, LB_T = LB_PIV + LB_TH;
static_assert(LB_T + LB_TH <= IPM_LB_SMALL, "record of the limited-memory update");

__device__ inline double lb_sum(double v, double* sh) {   // sum over the workgroup (fixed order), result in every thread
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (l == 0) sh[w] = v;
  __syncthreads();
  double s = 0.0;
  for (int q = 0; q < nw; ++q) s += sh[q];
  return s;
}

// column `a` (0 .. TH-1) of Q at variable i
__device__ inline double lb_q(const IpmDev& D, int bi, int a, int i, double sigma) {
  const size_t base = (size_t(bi) * IPM_LB_H + (a < IPM_LB_H ? a : a - IPM_LB_H)) * D.n + i;
  return a < IPM_LB_H ? sigma * D.lb_S[base] : D.lb_Y[base];
}

// one pair per accepted step, then the small matrix M; the current iterate becomes the previous one
__global__ __launch_bounds__(1024) void lb_update_kernel(IpmDev D) {
  __shared__ double sh[16];
  const int bi = blockIdx.y, t = threadIdx.x, nt = blockDim.x;
  const IpmInst& S = D.inst[bi];
  if (S.status != 0) return;
  double* rec = D.lb_small + size_t(bi) * IPM_LB_SMALL;
  if (S.mode != 0) {   // restoration phase / its multiplier pass: the point moves by another problem's steps -> empty memory
    if (t == 0) { rec[0] = 1.0; rec[1] = 0.0; rec[2] = 0.0; rec[3] = 0.0; }
    return;
  }
  const int n = D.n;
  const double *v = D.v + size_t(bi) * D.nv, *vl = D.vl + size_t(bi) * D.nv, *vu = D.vu + size_t(bi) * D.nv;
  const double *glag = D.glag + size_t(bi) * D.nv, *gold = D.lb_gold + size_t(bi) * D.nv;
  double* xprev = D.lb_xprev + size_t(bi) * n;
  double* Sm = D.lb_S + size_t(bi) * IPM_LB_H * n;
  double* Ym = D.lb_Y + size_t(bi) * IPM_LB_H * n;
  int c = int(rec[1]);
  double sigma = rec[0];
  const double skipped = rec[2];
  const bool has_prev = rec[3] != 0.0;
  __syncthreads();   // every thread has read the record before thread 0 changes it
  bool changed = false;
  if (has_prev) {
    double sts = 0, sty = 0, yty = 0;
    for (int i = t; i < n; i += nt) {
      const double s = v[i] - xprev[i];
      const double y = vl[i] != vu[i] ? glag[i] - gold[i] : 0.0;
      sts += s * s; sty += s * y; yty += y * y;
    }
    sts = lb_sum(sts, sh); sty = lb_sum(sty, sh); yty = lb_sum(yty, sh);
    if (sts > 0.0) {   // (a pass that only replaced lambda has s = 0: nothing to learn)
      if (sty > 1.4901161193847656e-08 * sqrt(sts) * sqrt(yty)) {
        if (c == IPM_LB_H) {
          for (int i = t; i < n; i += nt)
            for (int k = 0; k + 1 < IPM_LB_H; ++k) { Sm[size_t(k) * n + i] = Sm[size_t(k + 1) * n + i]; Ym[size_t(k) * n + i] = Ym[size_t(k + 1) * n + i]; }
          c = IPM_LB_H - 1;
        }
        for (int i = t; i < n; i += nt) {
          Sm[size_t(c) * n + i] = v[i] - xprev[i];
          Ym[size_t(c) * n + i] = vl[i] != vu[i] ? glag[i] - gold[i] : 0.0;
        }
        c += 1;
        sigma = fmin(1e8, fmax(1e-8, sty / sts));
        if (t == 0) { rec[0] = sigma; rec[1] = double(c); rec[2] = 0.0; rec[4] += 1.0; }
        changed = true;
      } else {
        const double sk = skipped + 1.0;
        if (sk >= 2.0) { c = 0; sigma = 1.0; changed = true; }
        if (t == 0) { rec[5] += 1.0; rec[2] = sk >= 2.0 ? 0.0 : sk; if (sk >= 2.0) { rec[0] = 1.0; rec[1] = 0.0; } }
      }
    }
  }
  __syncthreads();
  if (changed) {   // M = [[sigma S'S, L], [L', -D]] in the fixed layout, identity on the unused indices
    double* M = rec + LB_M;
    for (int q = t; q < LB_TH * LB_TH; q += nt) M[q] = (q / LB_TH == q % LB_TH) ? 1.0 : 0.0;
    __syncthreads();
    for (int a = 0; a < c; ++a)
      for (int b = 0; b <= a; ++b) {
        double ss = 0, sy = 0, ys = 0;
        for (int i = t; i < n; i += nt) {
          const double sa = Sm[size_t(a) * n + i], sb = Sm[size_t(b) * n + i];
          ss += sa * sb; sy += sa * Ym[size_t(b) * n + i]; ys += Ym[size_t(a) * n + i] * sb;
        }
        ss = lb_sum(ss, sh); sy = lb_sum(sy, sh); ys = lb_sum(ys, sh);
        if (t == 0) {
          M[a * LB_TH + b] = M[b * LB_TH + a] = sigma * ss;
          if (a == b) {
            M[(IPM_LB_H + a) * LB_TH + IPM_LB_H + a] = -sy;                  // -D
          } else {                                                            // L(a, b) = s_a'y_b for a > b (strictly lower)
            M[a * LB_TH + IPM_LB_H + b] = M[(IPM_LB_H + b) * LB_TH + a] = sy;
            M[b * LB_TH + IPM_LB_H + a] = M[(IPM_LB_H + a) * LB_TH + b] = 0.0;
            (void)ys;
          }
        }
      }
  }
  __syncthreads();
  for (int i = t; i < n; i += nt) xprev[i] = v[i];
  if (t == 0) rec[3] = 1.0;
}

// Z_j <- column j of E (Q at the positions of x, zero elsewhere), to be solved in place; blockIdx.x = column
__global__ __launch_bounds__(1024) void lb_column_kernel(IpmDev D) {
  const int bi = blockIdx.y, t = threadIdx.x, nt = blockDim.x, j = blockIdx.x;
  const IpmInst& S = D.inst[bi];
  if (S.status != 0) return;
  double* z = D.lb_Z + (size_t(j) * D.B + bi) * D.Nt;
  for (int p = t; p < D.Nt; p += nt) z[p] = 0.0;
  const double* rec = D.lb_small + size_t(bi) * IPM_LB_SMALL;
  const int c = int(rec[1]), a = j < IPM_LB_H ? j : j - IPM_LB_H;
  if (S.mode != 0 || a >= c) return;
  __syncthreads();
  const double sigma = rec[0];
  for (int i = t; i < D.n; i += nt) z[D.pos[i]] = lb_q(D, bi, j, i, sigma);
}

// C = M - E'Z and its LU factorisation (partial pivoting; 2 x history rows)
__global__ __launch_bounds__(1024) void lb_small_kernel(IpmDev D) {
  __shared__ double sh[16];
  const int bi = blockIdx.y, t = threadIdx.x, nt = blockDim.x;
  const IpmInst& S = D.inst[bi];
  if (S.status != 0 || S.mode != 0) return;
  double* rec = D.lb_small + size_t(bi) * IPM_LB_SMALL;
  const int c = int(rec[1]);
  if (c == 0) return;
  const double sigma = rec[0];
  double* C = rec + LB_C;
  const double* M = rec + LB_M;
  for (int q = t; q < LB_TH * LB_TH; q += nt) C[q] = M[q];
  __syncthreads();
  for (int ia = 0; ia < 2 * c; ++ia)
    for (int ib = ia; ib < 2 * c; ++ib) {
      const int a = ia < c ? ia : IPM_LB_H + ia - c, b = ib < c ? ib : IPM_LB_H + ib - c;
      const double* zb = D.lb_Z + (size_t(b) * D.B + bi) * D.Nt;
      double acc = 0.0;
      for (int i = t; i < D.n; i += nt) acc += lb_q(D, bi, a, i, sigma) * zb[D.pos[i]];
      acc = lb_sum(acc, sh);
      if (t == 0) {
        C[a * LB_TH + b] -= acc;
        if (a != b) C[b * LB_TH + a] -= acc;    // K0 is symmetric: E'K0^-1E is
      }
    }
  __syncthreads();
  if (t == 0) {
    double* piv = rec + LB_PIV;
    for (int k = 0; k < LB_TH; ++k) {
      int p = k;
      for (int r = k + 1; r < LB_TH; ++r)
        if (fabs(C[r * LB_TH + k]) > fabs(C[p * LB_TH + k])) p = r;
      piv[k] = double(p);
      if (p != k)
        for (int q = 0; q < LB_TH; ++q) { const double w = C[k * LB_TH + q]; C[k * LB_TH + q] = C[p * LB_TH + q]; C[p * LB_TH + q] = w; }
      const double d = C[k * LB_TH + k];
      for (int r = k + 1; r < LB_TH; ++r) {
        const double f = C[r * LB_TH + k] / d;
        C[r * LB_TH + k] = f;
        for (int q = k + 1; q < LB_TH; ++q) C[r * LB_TH + q] -= f * C[k * LB_TH + q];
      }
    }
  }
}

// d <- d + Z (M - E'Z)^-1 E'd for the solution d of K0 d = r that sits in rhs
__global__ __launch_bounds__(1024) void lb_correct_kernel(IpmDev D, double* rhs_all, int check_status) {
  __shared__ double sh[16];
  __shared__ double w[LB_TH];
  const int bi = blockIdx.y, t = threadIdx.x, nt = blockDim.x;
  const IpmInst& S = D.inst[bi];
  if (S.status != 0 || S.mode != 0 || (check_status == 2 && !S.soc_req)) return;
  double* rec = D.lb_small + size_t(bi) * IPM_LB_SMALL;
  const int c = int(rec[1]);
  if (c == 0) return;
  const double sigma = rec[0];
  double* d = rhs_all + size_t(bi) * D.Nt;
  if (t < LB_TH) w[t] = 0.0;
  __syncthreads();
  for (int ia = 0; ia < 2 * c; ++ia) {
    const int a = ia < c ? ia : IPM_LB_H + ia - c;
    double acc = 0.0;
    for (int i = t; i < D.n; i += nt) acc += lb_q(D, bi, a, i, sigma) * d[D.pos[i]];
    acc = lb_sum(acc, sh);
    if (t == 0) w[a] = acc;
  }
  __syncthreads();
  if (t == 0) {   // w <- C^-1 w with the LU factors
    const double *C = rec + LB_C, *piv = rec + LB_PIV;
    for (int k = 0; k < LB_TH; ++k) {   // the row interchanges first (whole rows were swapped, multipliers included), then L, then U
      const int p = int(piv[k]);
      if (p != k) { const double x = w[k]; w[k] = w[p]; w[p] = x; }
    }
    for (int k = 0; k < LB_TH; ++k)
      for (int r = k + 1; r < LB_TH; ++r) w[r] -= C[r * LB_TH + k] * w[k];
    for (int k = LB_TH - 1; k >= 0; --k) {
      double x = w[k];
      for (int q = k + 1; q < LB_TH; ++q) x -= C[k * LB_TH + q] * w[q];
      w[k] = x / C[k * LB_TH + k];
    }
  }
  __syncthreads();
  for (int p = t; p < D.Nt; p += nt) {
    double acc = 0.0;
    for (int ia = 0; ia < 2 * c; ++ia) {
      const int a = ia < c ? ia : IPM_LB_H + ia - c;
      acc += D.lb_Z[(size_t(a) * D.B + bi) * D.Nt + p] * w[a];
    }
    d[p] += acc;
  }
}

__global__ void lb_reset_kernel(IpmDev D) {
  const int bi = blockIdx.x * blockDim.x + threadIdx.x;
  if (bi >= D.B) return;
  double* rec = D.lb_small + size_t(bi) * IPM_LB_SMALL;
  for (int q = 0; q < 8; ++q) rec[q] = 0.0;
  rec[0] = 1.0;
}

void lb_launch_reset(const IpmDev& D, hipStream_t st) {
  hipLaunchKernelGGL(lb_reset_kernel, dim3(unsigned((D.B + 255) / 256)), dim3(256), 0, st, D);
}
void lb_launch_update(const IpmDev& D, hipStream_t st) {
  hipLaunchKernelGGL(lb_update_kernel, dim3(1, unsigned(D.B)), dim3(D.n >= 4096 ? 1024 : 256), 0, st, D);
}
// Z = K0^-1 E: all 2 x history columns of every running instance in ONE pass of the substitution kernels (the columns of an
// instance are right-hand sides j * B + bi of the same factors, IpmDev::rhs_mult)
void lb_launch_columns_and_solve(const IpmDev& D, hipStream_t st) {
  hipLaunchKernelGGL(lb_column_kernel, dim3(unsigned(2 * IPM_LB_H), unsigned(D.B)), dim3(D.Nt >= 4096 ? 1024 : 256), 0, st, D);
  IpmDev Dz = D;
  Dz.rhs = D.lb_Z;
  Dz.rhs_mult = 2 * IPM_LB_H;
  kkt_launch_solve(Dz, 1, st);
}
void lb_launch_small(const IpmDev& D, hipStream_t st) {
  hipLaunchKernelGGL(lb_small_kernel, dim3(1, unsigned(D.B)), dim3(D.n >= 4096 ? 1024 : 256), 0, st, D);
}
void lb_launch_correct(const IpmDev& D, int check_status, hipStream_t st) {
  hipLaunchKernelGGL(lb_correct_kernel, dim3(1, unsigned(D.B)), dim3(D.Nt >= 4096 ? 1024 : 256), 0, st, D, D.rhs, check_status);
}

}  // namespace rpm

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

// A sum over the variables is formed as a single workgroup of nt threads would form it — thread t adds its terms i = t, t + nt, ...
// in order, the 64 threads of a wave are folded by shuffles, the nt / 64 waves' sums are added in order — but every wave of that
// layout is a task of its own (task = sum * NW + wave; 4 tasks per workgroup of 256), so that the 144 sums of C = M - E'Z over
// 40 000 variables run on all CUs instead of one (metric problem: 2.0 ms -> tens of microseconds per iteration, same bits).  The
// waves' sums wait in lb_part (per instance LB_PART doubles: sum k at [k * 16, k * 16 + NW)); lb_fold adds them.
constexpr int LB_TASKS = IPM_LB_PART / 16, LB_PART = IPM_LB_PART;
static_assert(LB_TH * (LB_TH + 1) / 2 <= LB_TASKS, "sums of one phase");
template <class F>
__device__ inline void lb_wave_sum(int w, int nt, int n, F term, double* out) {
  const int l = threadIdx.x & 63;
  double v = 0.0;
  for (int i = 64 * w + l; i < n; i += nt) v += term(i);
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if (l == 0) *out = v;
}
__device__ inline double lb_fold(const double* part, int nw) {
  double s = 0.0;
  for (int q = 0; q < nw; ++q) s += part[q];
  return s;
}
// task of this wave: sum k, wave w of the layout; false beyond the n_sums sums of the phase
__device__ inline bool lb_task(int n_sums, int nw, int* k, int* w) {
  const int task = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  *k = task / nw;
  *w = task % nw;
  return *k < n_sums;
}

// column `a` (0 .. TH-1) of Q at variable i
__device__ inline double lb_q(const IpmDev& D, int bi, int a, int i, double sigma) {
  const size_t base = (size_t(bi) * IPM_LB_H + (a < IPM_LB_H ? a : a - IPM_LB_H)) * D.n + i;
  return a < IPM_LB_H ? sigma * D.lb_S[base] : D.lb_Y[base];
}

// ---- one pair per accepted step, then the small matrix M; the current iterate becomes the previous one ----
// record entries 6, 7: what the decision asks of the kernels behind it (6: 1 = store the pair in column rec[7] (after shifting the
// columns down when rec[7] = H - 1 and the memory was full: rec[7] + 16), 2 = memory emptied; 0 = nothing)
__global__ __launch_bounds__(256) void lb_stats_kernel(IpmDev D, int nw) {        // s's, s'y, y'y of the step just taken
  const int bi = blockIdx.y;
  const IpmInst& S = D.inst[bi];
  if (S.status != 0 || S.mode != 0) return;
  const double* rec = D.lb_small + size_t(bi) * IPM_LB_SMALL;
  if (rec[3] == 0.0) return;
  int k, w;
  if (!lb_task(3, nw, &k, &w)) return;
  const double *v = D.v + size_t(bi) * D.nv, *vl = D.vl + size_t(bi) * D.nv, *vu = D.vu + size_t(bi) * D.nv;
  const double *glag = D.glag + size_t(bi) * D.nv, *gold = D.lb_gold + size_t(bi) * D.nv, *xprev = D.lb_xprev + size_t(bi) * D.n;
  double* out = D.lb_part + size_t(bi) * LB_PART + k * 16 + w;
  const int nt = 64 * nw;
  if (k == 0) lb_wave_sum(w, nt, D.n, [&](int i) { const double s = v[i] - xprev[i]; return s * s; }, out);
  else if (k == 1) lb_wave_sum(w, nt, D.n, [&](int i) { const double s = v[i] - xprev[i]; const double y = vl[i] != vu[i] ? glag[i] - gold[i] : 0.0; return s * y; }, out);
  else lb_wave_sum(w, nt, D.n, [&](int i) { const double y = vl[i] != vu[i] ? glag[i] - gold[i] : 0.0; return y * y; }, out);
}
__global__ void lb_decide_kernel(IpmDev D, int nw) {
  const int bi = blockIdx.x * blockDim.x + threadIdx.x;
  if (bi >= D.B) return;
  const IpmInst& S = D.inst[bi];
  if (S.status != 0) return;
  double* rec = D.lb_small + size_t(bi) * IPM_LB_SMALL;
  rec[6] = 0.0;
  if (S.mode != 0) {   // restoration phase / its multiplier pass: the point moves by another problem's steps -> empty memory
    rec[0] = 1.0; rec[1] = 0.0; rec[2] = 0.0; rec[3] = 0.0;
    return;
  }
  int c = int(rec[1]);
  const double skipped = rec[2];
  if (rec[3] != 0.0) {
    const double* part = D.lb_part + size_t(bi) * LB_PART;
    const double sts = lb_fold(part, nw), sty = lb_fold(part + 16, nw), yty = lb_fold(part + 32, nw);
    if (sts > 0.0) {   // (a pass that only replaced lambda has s = 0: nothing to learn)
      if (sty > 1.4901161193847656e-08 * sqrt(sts) * sqrt(yty)) {
        double col = double(c);
        if (c == IPM_LB_H) { c = IPM_LB_H - 1; col = double(c) + 16.0; }
        rec[6] = 1.0; rec[7] = col;
        c += 1;
        rec[0] = fmin(1e8, fmax(1e-8, sty / sts)); rec[1] = double(c); rec[2] = 0.0; rec[4] += 1.0;
      } else {
        const double sk = skipped + 1.0;
        rec[5] += 1.0;
        rec[2] = sk >= 2.0 ? 0.0 : sk;
        if (sk >= 2.0) { rec[0] = 1.0; rec[1] = 0.0; rec[6] = 2.0; }
      }
    }
  }
  rec[3] = 1.0;
}
__global__ __launch_bounds__(256) void lb_store_kernel(IpmDev D) {      // the pair into its column; x becomes the previous iterate
  const int bi = blockIdx.y;
  const IpmInst& S = D.inst[bi];
  if (S.status != 0 || S.mode != 0) return;
  const double* rec = D.lb_small + size_t(bi) * IPM_LB_SMALL;
  const int n = D.n;
  const double *v = D.v + size_t(bi) * D.nv, *vl = D.vl + size_t(bi) * D.nv, *vu = D.vu + size_t(bi) * D.nv;
  const double *glag = D.glag + size_t(bi) * D.nv, *gold = D.lb_gold + size_t(bi) * D.nv;
  double* xprev = D.lb_xprev + size_t(bi) * n;
  double* Sm = D.lb_S + size_t(bi) * IPM_LB_H * n;
  double* Ym = D.lb_Y + size_t(bi) * IPM_LB_H * n;
  const bool store = rec[6] == 1.0;
  const bool shift = store && rec[7] >= 16.0;
  const int col = store ? int(rec[7]) & 15 : 0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (shift)
      for (int k = 0; k + 1 < IPM_LB_H; ++k) { Sm[size_t(k) * n + i] = Sm[size_t(k + 1) * n + i]; Ym[size_t(k) * n + i] = Ym[size_t(k + 1) * n + i]; }
    if (store) {
      Sm[size_t(col) * n + i] = v[i] - xprev[i];
      Ym[size_t(col) * n + i] = vl[i] != vu[i] ? glag[i] - gold[i] : 0.0;
    }
    xprev[i] = v[i];
  }
}
// M = [[sigma S'S, L], [L', -D]]: s_a's_b and s_a'y_b for b <= a < pairs held (sum 2 (a (a + 1) / 2 + b) + {0, 1})
__global__ __launch_bounds__(256) void lb_mdots_kernel(IpmDev D, int nw) {
  const int bi = blockIdx.y;
  const IpmInst& S = D.inst[bi];
  if (S.status != 0 || S.mode != 0) return;
  const double* rec = D.lb_small + size_t(bi) * IPM_LB_SMALL;
  if (rec[6] == 0.0) return;
  const int c = int(rec[1]);
  int k, w;
  if (!lb_task(c * (c + 1), nw, &k, &w)) return;
  int a = 0;
  while ((a + 1) * (a + 2) / 2 <= k / 2) ++a;
  const int b = k / 2 - a * (a + 1) / 2;
  const int n = D.n;
  const double* sa = D.lb_S + (size_t(bi) * IPM_LB_H + a) * n;
  const double* other = (k & 1) ? D.lb_Y + (size_t(bi) * IPM_LB_H + b) * n : D.lb_S + (size_t(bi) * IPM_LB_H + b) * n;
  lb_wave_sum(w, 64 * nw, n, [&](int i) { return sa[i] * other[i]; }, D.lb_part + size_t(bi) * LB_PART + k * 16 + w);
}
__global__ __launch_bounds__(64) void lb_mfinish_kernel(IpmDev D, int nw) {      // one wave per instance: a lane per sum
  const int bi = blockIdx.x, t = threadIdx.x;
  const IpmInst& S = D.inst[bi];
  if (S.status != 0 || S.mode != 0) return;
  double* rec = D.lb_small + size_t(bi) * IPM_LB_SMALL;
  if (rec[6] == 0.0) return;
  const int c = int(rec[1]);
  const double sigma = rec[0];
  double* M = rec + LB_M;
  const double* part = D.lb_part + size_t(bi) * LB_PART;
  for (int q = t; q < LB_TH * LB_TH; q += 64) M[q] = (q / LB_TH == q % LB_TH) ? 1.0 : 0.0;     // identity on the unused indices
  __syncthreads();
  if (t < c * (c + 1) / 2) {
    int a = 0;
    while ((a + 1) * (a + 2) / 2 <= t) ++a;
    const int b = t - a * (a + 1) / 2, k = 2 * t;
    const double ss = lb_fold(part + k * 16, nw), sy = lb_fold(part + (k + 1) * 16, nw);
    M[a * LB_TH + b] = M[b * LB_TH + a] = sigma * ss;
    if (a == b) {
      M[(IPM_LB_H + a) * LB_TH + IPM_LB_H + a] = -sy;                  // -D
    } else {                                                            // L(a, b) = s_a'y_b for a > b (strictly lower)
      M[a * LB_TH + IPM_LB_H + b] = M[(IPM_LB_H + b) * LB_TH + a] = sy;
      M[b * LB_TH + IPM_LB_H + a] = M[(IPM_LB_H + a) * LB_TH + b] = 0.0;
    }
  }
}

// Z_j <- column j of E (Q at the positions of x, zero elsewhere — the launcher has zeroed Z), to be solved in place;
// blockIdx.x = column + 2 H x slice of the variables
__global__ __launch_bounds__(256) void lb_column_kernel(IpmDev D) {
  const int bi = blockIdx.y, j = blockIdx.x % LB_TH, slice = blockIdx.x / LB_TH, n_slices = gridDim.x / LB_TH;
  const IpmInst& S = D.inst[bi];
  if (S.status != 0) return;
  const double* rec = D.lb_small + size_t(bi) * IPM_LB_SMALL;
  const int c = int(rec[1]), a = j < IPM_LB_H ? j : j - IPM_LB_H;
  if (S.mode != 0 || a >= c) return;
  double* z = D.lb_Z + (size_t(j) * D.B + bi) * D.Nt;
  const double sigma = rec[0];
  for (int i = slice * blockDim.x + threadIdx.x; i < D.n; i += n_slices * blockDim.x) z[D.pos[i]] = lb_q(D, bi, j, i, sigma);
}

// C = M - E'Z and its LU factorisation (partial pivoting; 2 x history rows): the sums q_a'z_b, a <= b (sum ia (2 TH - ia + 1) / 2 + ib - ia
// in the packed numbering of the live columns), then one thread per instance
__device__ inline void lb_pair_of(int k, int m, int* ia, int* ib) {   // k-th pair (ia <= ib < m), rows first
  int r = 0;
  while (k >= m - r) { k -= m - r; ++r; }
  *ia = r;
  *ib = r + k;
}
__global__ __launch_bounds__(256) void lb_cdots_kernel(IpmDev D, int nw) {
  const int bi = blockIdx.y;
  const IpmInst& S = D.inst[bi];
  if (S.status != 0 || S.mode != 0) return;
  const double* rec = D.lb_small + size_t(bi) * IPM_LB_SMALL;
  const int c = int(rec[1]);
  if (c == 0) return;
  int k, w;
  if (!lb_task(c * (2 * c + 1), nw, &k, &w)) return;
  int ia, ib;
  lb_pair_of(k, 2 * c, &ia, &ib);
  const int a = ia < c ? ia : IPM_LB_H + ia - c, b = ib < c ? ib : IPM_LB_H + ib - c;
  const double sigma = rec[0];
  const double* zb = D.lb_Z + (size_t(b) * D.B + bi) * D.Nt;
  lb_wave_sum(w, 64 * nw, D.n, [&](int i) { return lb_q(D, bi, a, i, sigma) * zb[D.pos[i]]; }, D.lb_part + size_t(bi) * LB_PART + k * 16 + w);
}
__global__ __launch_bounds__(LB_TH * LB_TH) void lb_cfinish_kernel(IpmDev D, int nw) {   // one workgroup per instance: a thread per entry of C
  __shared__ double Cs[LB_TH * LB_TH];
  __shared__ int pv;
  const int bi = blockIdx.x, t = threadIdx.x, r = t / LB_TH, q = t % LB_TH;
  const IpmInst& S = D.inst[bi];
  if (S.status != 0 || S.mode != 0) return;
  double* rec = D.lb_small + size_t(bi) * IPM_LB_SMALL;
  const int c = int(rec[1]);
  if (c == 0) return;
  const double* part = D.lb_part + size_t(bi) * LB_PART;
  Cs[t] = rec[LB_M + t];
  __syncthreads();
  if (t < c * (2 * c + 1)) {
    int ia, ib;
    lb_pair_of(t, 2 * c, &ia, &ib);
    const int a = ia < c ? ia : IPM_LB_H + ia - c, b = ib < c ? ib : IPM_LB_H + ib - c;
    const double acc = lb_fold(part + t * 16, nw);
    Cs[a * LB_TH + b] -= acc;
    if (a != b) Cs[b * LB_TH + a] -= acc;    // K0 is symmetric: E'K0^-1E is
  }
  __syncthreads();
  // LU with partial pivoting, the first largest entry of a column as the pivot; every entry sees the operations of the
  // one-thread elimination in the same order
  for (int k = 0; k < LB_TH; ++k) {
    if (t == 0) {
      int p = k;
      for (int r2 = k + 1; r2 < LB_TH; ++r2)
        if (fabs(Cs[r2 * LB_TH + k]) > fabs(Cs[p * LB_TH + k])) p = r2;
      pv = p;
      rec[LB_PIV + k] = double(p);
    }
    __syncthreads();
    const int p = pv;
    const bool swap = p != k && (r == k || r == p);      // whole rows change places
    double other = 0.0;
    if (swap) other = Cs[(r == k ? p : k) * LB_TH + q];
    __syncthreads();
    if (swap) Cs[t] = other;
    __syncthreads();
    const double d = Cs[k * LB_TH + k];
    double f = 0.0;
    if (r > k) f = Cs[r * LB_TH + k] / d;
    __syncthreads();
    if (r > k) {
      if (q == k) Cs[t] = f;
      else if (q > k) Cs[t] -= f * Cs[k * LB_TH + q];
    }
    __syncthreads();
  }
  rec[LB_C + t] = Cs[t];
}

// d <- d + Z (M - E'Z)^-1 E'd for the solution d of K0 d = r that sits in rhs: the sums q_a'd first ...
__global__ __launch_bounds__(256) void lb_wdots_kernel(IpmDev D, const double* rhs_all, int check_status, int nw) {
  const int bi = blockIdx.y;
  const IpmInst& S = D.inst[bi];
  if (S.status != 0 || S.mode != 0 || (check_status == 2 && !S.soc_req)) return;
  const double* rec = D.lb_small + size_t(bi) * IPM_LB_SMALL;
  const int c = int(rec[1]);
  if (c == 0) return;
  int ia, w;
  if (!lb_task(2 * c, nw, &ia, &w)) return;
  const int a = ia < c ? ia : IPM_LB_H + ia - c;
  const double sigma = rec[0];
  const double* d = rhs_all + size_t(bi) * D.Nt;
  lb_wave_sum(w, 64 * nw, D.n, [&](int i) { return lb_q(D, bi, a, i, sigma) * d[D.pos[i]]; }, D.lb_part + size_t(bi) * LB_PART + ia * 16 + w);
}
// ... then every workgroup solves the 12 x 12 system for itself (the same operations in each) and corrects its slice of d
__global__ __launch_bounds__(256) void lb_apply_kernel(IpmDev D, double* rhs_all, int check_status, int nw) {
  __shared__ double w[LB_TH];
  const int bi = blockIdx.y, t = threadIdx.x;
  const IpmInst& S = D.inst[bi];
  if (S.status != 0 || S.mode != 0 || (check_status == 2 && !S.soc_req)) return;
  const double* rec = D.lb_small + size_t(bi) * IPM_LB_SMALL;
  const int c = int(rec[1]);
  if (c == 0) return;
  double* d = rhs_all + size_t(bi) * D.Nt;
  if (t == 0) {
    const double* part = D.lb_part + size_t(bi) * LB_PART;
    for (int q = 0; q < LB_TH; ++q) w[q] = 0.0;
    for (int ia = 0; ia < 2 * c; ++ia) w[ia < c ? ia : IPM_LB_H + ia - c] = lb_fold(part + ia * 16, nw);
    // w <- C^-1 w with the LU factors
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
  for (int p = blockIdx.x * blockDim.x + t; p < D.Nt; p += gridDim.x * blockDim.x) {
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
static unsigned lb_task_blocks(int n_sums, int nw) { return unsigned((n_sums * nw + 3) / 4); }
static int lb_waves(int len) { return len >= 4096 ? 16 : 4; }    // the layout the sums are defined on: 1024 resp. 256 threads
void lb_launch_update(const IpmDev& D, hipStream_t st) {
  const int nw = lb_waves(D.n);
  const unsigned B = unsigned(D.B), ib = unsigned((D.B + 63) / 64), vb = unsigned(std::max(1, std::min(1024, (D.n + 255) / 256)));
  hipLaunchKernelGGL(lb_stats_kernel, dim3(lb_task_blocks(3, nw), B), dim3(256), 0, st, D, nw);
  hipLaunchKernelGGL(lb_decide_kernel, dim3(ib), dim3(64), 0, st, D, nw);
  hipLaunchKernelGGL(lb_store_kernel, dim3(vb, B), dim3(256), 0, st, D);
  hipLaunchKernelGGL(lb_mdots_kernel, dim3(lb_task_blocks(IPM_LB_H * (IPM_LB_H + 1), nw), B), dim3(256), 0, st, D, nw);
  hipLaunchKernelGGL(lb_mfinish_kernel, dim3(B), dim3(64), 0, st, D, nw);
}
// Z = K0^-1 E: all 2 x history columns of every running instance in ONE pass of the substitution kernels (the columns of an
// instance are right-hand sides j * B + bi of the same factors, IpmDev::rhs_mult)
void lb_launch_columns_and_solve(const IpmDev& D, hipStream_t st) {
  (void)hipMemsetAsync(D.lb_Z, 0, size_t(2 * IPM_LB_H) * D.B * D.Nt * sizeof(double), st);
  const int slices = std::max(1, std::min(64, (D.n + 1023) / 1024));
  hipLaunchKernelGGL(lb_column_kernel, dim3(unsigned(2 * IPM_LB_H * slices), unsigned(D.B)), dim3(256), 0, st, D);
  IpmDev Dz = D;
  Dz.rhs = D.lb_Z;
  Dz.rhs_mult = 2 * IPM_LB_H;
  kkt_launch_solve(Dz, 1, st);
}
void lb_launch_small(const IpmDev& D, hipStream_t st) {
  const int nw = lb_waves(D.n);
  hipLaunchKernelGGL(lb_cdots_kernel, dim3(lb_task_blocks(LB_TH * (LB_TH + 1) / 2, nw), unsigned(D.B)), dim3(256), 0, st, D, nw);
  hipLaunchKernelGGL(lb_cfinish_kernel, dim3(unsigned(D.B)), dim3(LB_TH * LB_TH), 0, st, D, nw);
}
void lb_launch_correct(const IpmDev& D, int check_status, hipStream_t st) {
  const int nw = lb_waves(D.Nt);
  const unsigned vb = unsigned(std::max(1, std::min(1024, (D.Nt + 255) / 256)));
  hipLaunchKernelGGL(lb_wdots_kernel, dim3(lb_task_blocks(LB_TH, nw), unsigned(D.B)), dim3(256), 0, st, D, D.rhs, check_status, nw);
  hipLaunchKernelGGL(lb_apply_kernel, dim3(vb, unsigned(D.B)), dim3(256), 0, st, D, D.rhs, check_status, nw);
}

}  // namespace rpm

// rpm_device.hip — hand-written HIP kernels for gfx950 (MI355X / CDNA4) and the device half of
// the engine.  No CPU fallback lives here or anywhere else in the product.
//
// Kernels (DESIGN.md §Kernels has the roofline of each):
//   rpm_tile_kernel   K1+K2+K3 fused: per-node dynamics/path evaluation, forward-difference (or
//                     analytic) node Jacobian, LGR defect D.X - (dt/2) f, and the coalesced block
//                     scatter of the COO Jacobian values, plus this workgroup's share of the
//                     constant Doffdiag block.  Replaces NLPWrapper::GetConsFun (LpNLPWrapper.cpp:55-229),
//                     GetPhaseJacbi (:524-862), LpFDderive::DerivDae (LpFiniteDifferenceDerive.cpp:194-324)
//                     and dsmatrix::operator* (SparseMatrix/LpSparseMatrix.cpp:127-155).
//   endpoint block    K4: events, linkages, A_lin.x rows and their Jacobian entries
//                     (LpNLPWrapper.cpp:125-136,180-211,406-522,833-861; :45,:242); one extra workgroup
//                     of the same launch.
//   rpm_obj_kernel    K5: objective quadrature and gradient (GetObjFun :863-939, GetObjGrad :940-1104).
//
// Thread layout of rpm_tile_kernel: a workgroup owns a tile of <= T consecutive collocation nodes of one
// phase; thread = (role, node) with node fastest, so that the N-long diagonal runs of every Jacobian
// block are written by consecutive lanes (coalesced 8-byte stores).  Role 0 evaluates the unperturbed
// dynamics, role 1+v the dynamics with variable v perturbed (v = states, controls, time) — the
// reference's (2+nx+nu) whole-vector user calls become (2+nx+nu) roles evaluated concurrently.
// State roles also compute their state's D.X row from the LDS-staged D rows and X tile.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <utility>
#include <vector>

#include "problems/problems.hpp"
#include "rpm_engine.hpp"

namespace rpm {

// ------------------------------------------------------------------------------------------
struct KParams {
  const PhaseDev* phases;
  const TileDev* tiles;  // the tiles this rank computes, compact
  int n_my_tiles;
  const TaskDev* tasks;  // endpoint work items, one extra workgroup each
  int n_tasks;
  const NodeDev* nodes;
  const double* points;
  const double* weights;
  const double* diag;
  const double* dvals;
  const double* doff_vals;
  const double* consts;
  const LinkDev* links;
  const int* alin_j;     // 2 entries per linear row
  const double* alin_v;
  double tol;
  int P, L, n, m, m_nl, nnz, nnz_nl, nnz_lin, nnz_const;
  int max_span, max_drow;
  int max_cshare;                       // largest constant-block share of a tile (c_cnt)
  int diag_mask;                        // ablation mask, only honoured by the -DRPM_DIAG diagnostic build
  unsigned long long* trace;            // per-workgroup timestamps (diagnostic build with RPM_DIAG_TRACE set), else NULL
};

struct HParams {
  const HessPairDev* pairs;
  const HessPhaseDev* phases;
  const HessEndDev* ends;
  const HessLinkDev* links;
  const int* tiles;     // per workgroup: phase, k0, cnt
  int n_tiles, th, n_ends, n_links, nnz_h, tmp_len;
};

struct Device {
  int device_id = -1;
  int* d_flags2 = nullptr;      // two non-finite flag words (g, Jacobian) and their page-locked host mirror
  int* h_flags2 = nullptr;
  size_t trace_words = 0;
  hipStream_t stream = nullptr;
  KParams kp{};
  // tables
  PhaseDev* d_phases = nullptr;
  TileDev* d_tiles = nullptr;
  TaskDev* d_tasks = nullptr;
  NodeDev* d_nodes = nullptr;
  double *d_points = nullptr, *d_weights = nullptr, *d_diag = nullptr, *d_dvals = nullptr,
         *d_doff = nullptr, *d_consts = nullptr, *d_alin_v = nullptr;
  LinkDev* d_links = nullptr;
  int* d_alin_j = nullptr;
  // staging buffers of the host-pointer TNLP path
  double *d_x = nullptr, *d_g = nullptr, *d_values = nullptr, *d_grad = nullptr, *d_obj = nullptr,
         *d_lambda = nullptr, *d_hess = nullptr;
  double* d_partial = nullptr;  // objective partial sums
  int* d_flag = nullptr;        // non-finite flag of the host-pointer path
  std::vector<std::pair<const void*, size_t>> pinned;   // caller buffers registered with hipHostRegister
  bool cache_valid = false;     // d_g / d_values hold the pair of the x last uploaded
  size_t lds_bytes = 0;
  int pl_slots = 0;             // resident workgroups the pipelined kernel is launched with (2 per CU)
  size_t pl_lds = 0;
  bool pl_ok = false;           // the mesh fits rpm_tile_pl_kernel's register staging
  // exact-Hessian tables
  HessPairDev* d_hpairs = nullptr;
  HessPhaseDev* d_hphases = nullptr;
  HessEndDev* d_hends = nullptr;
  HessLinkDev* d_hlinks = nullptr;
  int* d_htiles = nullptr;
  double* d_htmp = nullptr;
  HParams hp{};
  size_t hess_lds = 0;
  int hess_threads = 0;
  struct SegTable { void* ptr = nullptr; int count = 0; int stride = -1; };
  SegTable segtab[2][2];        // [g|values][pack|unpack] run tables of the interval sharding
};

#define HIP_TRY(e, call)                                                                   \
  do {                                                                                     \
    hipError_t _s = (call);                                                                \
    if (_s != hipSuccess) {                                                                \
      (e).err = std::string(#call) + ": " + hipGetErrorString(_s);                         \
      return RPM_E_DEVICE;                                                                 \
    }                                                                                      \
  } while (0)

// ------------------------------------------------------------------------------------------
// endpoint rows: events, linkages, linear rows.  Each work item (TaskDev) is one workgroup of the same
// launch, so the three kinds run concurrently on different CUs.
// WAVE = true: the work item is done by ONE wave (lanes = perturbations, the base values travel by lane shuffle, no
// workgroup barrier), so a wave of a workgroup that is busy with something else can take it (rpm_tile_pl_kernel).
template <class Prob, bool WG, bool WJ, bool AN, bool WAVE = false>
__device__ void endpoint_block(const KParams& K, const TaskDev task, const double* __restrict__ x,
                               double* __restrict__ g, double* __restrict__ vals, double* lds) {
  constexpr int NX = Prob::NX;
  constexpr int NE = Prob::NE_MAX > 0 ? Prob::NE_MAX : 1;
  constexpr int NL = Prob::NLINK_MAX > 0 ? Prob::NLINK_MAX : 1;
  static_assert(!WAVE || 2 * NX + 3 <= 64, "endpoint perturbations must fit one wave");
  const int tid = WAVE ? int(threadIdx.x & 63) : int(threadIdx.x);
  const int nthr = WAVE ? 64 : int(blockDim.x);
  const double* c = K.consts;
  if (task.type == 0) {
    // linear rows  A_lin * x  (LpNLPWrapper.cpp:45; COO loop order of LpSparseMatrix.cpp:142-153) and
    // their constant Jacobian entries (:242)
    for (int r = tid; r < K.P + K.L; r += nthr) {
      if (WG) {
        double acc = 0.0;
        acc += K.alin_v[2 * r] * x[K.alin_j[2 * r]];
        acc += K.alin_v[2 * r + 1] * x[K.alin_j[2 * r + 1]];
        g[K.m_nl + r] = acc;
      }
      if (WJ) {
        vals[K.nnz_nl + 2 * r] = K.alin_v[2 * r];
        vals[K.nnz_nl + 2 * r + 1] = K.alin_v[2 * r + 1];
      }
    }
  } else if (task.type == 1) {
    // ---- events of one phase: lane 0 = base, lanes 1..2NX+2 = perturbations [x0.., t0, xf.., tf]
    //      (LpFDderive::DerivEvent, LpFiniteDifferenceDerive.cpp:326-409)
    const PhaseDev ph = K.phases[task.idx];
    const int pi = tid;
    const bool act = pi <= 2 * NX + 2;
    double x0[NX], xf[NX], ev[NE];
    double t0 = x[ph.x_t0], tf = x[ph.x_t0 + 1];
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      x0[j] = x[ph.x_state0 + j * (ph.N + 1)];
      xf[j] = x[ph.x_state0 + j * (ph.N + 1) + ph.N];
    }
    double h = 1.0;
    if (WJ && !AN && pi >= 1) {
      const int v = pi - 1;
#pragma unroll
      for (int j = 0; j < NX; ++j) {
        if (v == j) { h = K.tol * (fabs(x0[j]) + 1); x0[j] += h; }
        if (v == NX + 1 + j) { h = K.tol * (fabs(xf[j]) + 1); xf[j] += h; }
      }
      if (v == NX) { h = K.tol * (1 + fabs(t0)); t0 += h; }
      if (v == 2 * NX + 1) { h = K.tol * (1 + fabs(tf)); tf += h; }
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) ev[i] = 0.0;
    if (act && (pi == 0 || !AN)) Prob::event(ph.phase_num, t0, x0, tf, xf, c, ev);
    double base[NE];
    if constexpr (WAVE) {
#pragma unroll
      for (int i = 0; i < NE; ++i) base[i] = __shfl(ev[i], 0, 64);
    }
    if (pi == 0) {
#pragma unroll
      for (int i = 0; i < NE; ++i)
        if (i < ph.ne) {
          if (!WAVE) lds[i] = ev[i];
          if (WG) g[ph.g0 + (NX + Prob::NC) * ph.N + i] = ev[i];
        }
    }
    if constexpr (!WAVE) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NE; ++i) base[i] = lds[i];
    }
    if (WJ && act && pi >= 1) {
      const int v = pi - 1;
      double de[NE];
      if constexpr (AN) {
        Prob::event_jac_col(ph.phase_num, v, t0, x0, tf, xf, c, de);
      } else {
#pragma unroll
        for (int i = 0; i < NE; ++i) de[i] = (ev[i] - base[i]) / h;
      }
      // position inside an event's row of entries: (x0_j, xf_j) pairs, then t0, tf (:837-853)
      int pos;
      if (v < NX) pos = 2 * v;
      else if (v == NX) pos = 2 * NX;
      else if (v <= 2 * NX) pos = 2 * (v - NX - 1) + 1;
      else pos = 2 * NX + 1;
#pragma unroll
      for (int i = 0; i < NE; ++i)
        if (i < ph.ne) vals[ph.v_evt0 + i * (2 * NX + 2) + pos] = de[i];
    }
  } else {
    // ---- one linkage pair: lane 0 = base, 1..NX = xf_left perturbations, NX+1..2NX = x0_right
    //      (LpFDderive::DerivLink, LpFiniteDifferenceDerive.cpp:411-502)
    const LinkDev lk = K.links[task.idx];
    const PhaseDev pl = K.phases[lk.left];
    const PhaseDev pr = K.phases[lk.right];
    const int pi = tid;
    const bool act = pi <= 2 * NX;
    double xl[NX], xr[NX], lo[NL];
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      xl[j] = x[pl.x_state0 + j * (pl.N + 1) + pl.N];
      xr[j] = x[pr.x_state0 + j * (pr.N + 1)];
    }
    double h = 1.0;
    if (WJ && !AN && pi >= 1) {
      const int v = pi - 1;
#pragma unroll
      for (int j = 0; j < NX; ++j) {
        if (v == j) { h = K.tol * (1 + fabs(xl[j])); xl[j] += h; }
        if (v == NX + j) { h = K.tol * (1 + fabs(xr[j])); xr[j] += h; }
      }
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) lo[i] = 0.0;
    if (act && (pi == 0 || !AN)) Prob::link(lk.left + 1, lk.right + 1, xl, xr, c, lk.nlink, lo);
    double base[NL];
    if constexpr (WAVE) {
#pragma unroll
      for (int i = 0; i < NL; ++i) base[i] = __shfl(lo[i], 0, 64);
    }
    if (pi == 0) {
#pragma unroll
      for (int i = 0; i < NL; ++i)
        if (i < lk.nlink) {
          if (!WAVE) lds[i] = lo[i];
          if (WG) g[lk.g0 + i] = lo[i];
        }
    }
    if constexpr (!WAVE) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < NL; ++i) base[i] = lds[i];
    }
    if (WJ && act && pi >= 1) {
      const int v = pi - 1;
      double dl[NL];
      if constexpr (AN) {
        Prob::link_jac_col(lk.left + 1, lk.right + 1, v, xl, xr, c, lk.nlink, dl);
      } else {
#pragma unroll
        for (int i = 0; i < NL; ++i) dl[i] = (lo[i] - base[i]) / (1.0 * h);
      }
#pragma unroll
      for (int i = 0; i < NL; ++i)
        if (i < lk.nlink) vals[lk.v0 + v * lk.nlink + i] = dl[i];  // column-major, :461-501
    }
  }
}

// ------------------------------------------------------------------------------------------
template <class Prob, int T, bool WG, bool WJ, bool AN, bool DXM = false>
__global__ void rpm_tile_kernel(const KParams K, const double* __restrict__ xall,
                                double* __restrict__ gall, double* __restrict__ vall) {
  constexpr int NX = Prob::NX, NU = Prob::NU, NC = Prob::NC;
  constexpr int NO = NX + NC;              // outputs per node: f then c
  constexpr int NV = NX + NU + 1;          // perturbation variables: states, controls, time
  constexpr int NB = NX + NU + 2;          // Jacobian blocks per output row: x.., u.., t0, tf
  constexpr int R = WJ ? NV + 1 : (NX > 0 ? NX : 1);
  constexpr int NCs = NC > 0 ? NC : 1;
  extern __shared__ double lds[];
  const int tid = threadIdx.x, nthr = blockDim.x;
  const double* __restrict__ x = xall + size_t(blockIdx.y) * K.n;
  double* __restrict__ g = gall + size_t(blockIdx.y) * K.m;
  double* __restrict__ vals = vall + size_t(blockIdx.y) * K.nnz;
#ifdef RPM_DIAG
  if (K.diag_mask & 32) return;
  if ((K.diag_mask & 1) && int(blockIdx.x) >= K.n_my_tiles) return;
#endif
  if (int(blockIdx.x) >= K.n_my_tiles) {  // the launch's trailing workgroups: endpoint work items
    endpoint_block<Prob, WG, WJ, AN>(K, K.tasks[int(blockIdx.x) - K.n_my_tiles], x, g, vals, lds);
    return;
  }
  // XCD-aware tile order: workgroups b, b+8, b+16, ... are dealt to the same XCD, so give each XCD a
  // contiguous run of tiles; neighbouring 128-byte pieces of every Jacobian block then meet in one L2
  // and leave it as longer contiguous write-backs (speed only, correctness does not depend on placement)
  const int nt = K.n_my_tiles, per = nt >> 3, rem = nt & 7, xcd = int(blockIdx.x) & 7, slot = int(blockIdx.x) >> 3;
  const int tix = xcd * per + (xcd < rem ? xcd : rem) + slot;
  const TileDev tl = K.tiles[tix];
  const TileDev& ph = tl;   // the phase fields the kernel needs are replicated in the tile record
#ifdef RPM_DIAG
  if (K.diag_mask & 64) { if (tl.cnt < 0) vals[0] = 0; return; }
#endif
  const auto c = (const __attribute__((address_space(4))) double*)K.consts;   // constant address space: scalar loads
  double* Xs = lds;                          // [NX][max_span]  state-matrix rows the tile's D rows touch
  double* Us = Xs + NX * K.max_span;         // [NU][T]
  double* Ds = Us + NU * T;                  // the tile's D rows, row-major per node
  double* Fb = Ds + K.max_drow;              // [NO][T] unperturbed f and c
  double* DXs = Fb + NO * T;                 // [NX][T] D.X of the tile (MFMA variant only)

  // ---- issue the loads nothing depends on first: this thread's node record and its share of the
  //      constant-block sources (stored at the very end) ----
  const int kk = tid % T, role = tid / T;
  const int kc = kk < tl.cnt ? kk : tl.cnt - 1;   // clamp so idle lanes read valid memory
  const int k = tl.k0 + kc;
  const int nidx = ph.node0 + k;
  const double tau = K.points[nidx];
  const NodeDev nd = K.nodes[nidx];
  const double ddiag = WJ ? K.diag[nidx] : 0.0;
  constexpr int CPRE = 8;                          // constant-block sources prefetched per thread
  double cpre[CPRE > 0 ? CPRE : 1];
  if (WJ) {
#pragma unroll
    for (int u = 0; u < CPRE; ++u) {
      const int q = tid + u * nthr;
      cpre[u] = q < tl.c_cnt ? K.doff_vals[tl.c_src0 + q] : 0.0;
    }
  }

#ifdef RPM_DIAG
  if (K.diag_mask & 128) { if (tau + cpre[0] + cpre[5] + ddiag + nd.dlen == 1e300) vals[0] = 0; return; }
#endif
  // ---- stage X tile, U tile and D rows in LDS (coalesced: every run below is contiguous in HBM) ----
  for (int q = tid; q < NX * tl.span_len; q += nthr) {
    const int i = q / tl.span_len, r = q - i * tl.span_len;
    Xs[i * K.max_span + r] = x[ph.x_state0 + i * (ph.N + 1) + tl.span0 + r];
  }
  for (int q = tid; q < NU * tl.cnt; q += nthr) {
    const int j = q / tl.cnt, r = q - j * tl.cnt;
    Us[j * T + r] = x[ph.x_control0 + j * ph.N + tl.k0 + r];
  }
  if (WG)
    for (int q = tid; q < tl.drow_len; q += nthr) Ds[q] = K.dvals[tl.drow0 + q];
  const double t0 = x[ph.x_t0], tf = x[ph.x_t0 + 1];
#ifdef RPM_DIAG
  if (K.diag_mask & 256) { if (t0 == 1e300) vals[0] = 0; return; }
#endif
  __syncthreads();

#ifdef RPM_DIAG
  if (K.diag_mask & 16) return;
#endif
  const bool act = kk < tl.cnt && role < R;
  const double tspan = tf - t0;
  double tk = (tau + 1) * (tspan / 2.0) + t0;      // LpNLPWrapper.cpp:80
  double xs[NX > 0 ? NX : 1], us[NU > 0 ? NU : 1];
#pragma unroll
  for (int i = 0; i < NX; ++i) xs[i] = Xs[i * K.max_span + (k - tl.span0)];
#pragma unroll
  for (int j = 0; j < NU; ++j) us[j] = Us[j * T + kc];

  // ---- D.X for this thread's state: ascending-column sum, separate multiply and add, exactly the
  //      order of the reference's COO loop for one output row (LpSparseMatrix.cpp:142-153) ----
  const int sv = WJ ? role - 1 : role;
  double dx = 0.0;
  if (WG && !DXM && sv >= 0 && sv < NX) {
    const double* drow = Ds + (nd.drow_off - tl.drow0);
    const double* xcol = Xs + sv * K.max_span + (nd.dcol0 - tl.span0);
    for (int j = 0; j < nd.dlen; ++j) dx += drow[j] * xcol[j];
  }
  // ---- MFMA variant (dx_mode = 1): the tile's D.X as dense 16x16x4 FP64 matrix-core products.  The tile's D
  //      rows form a block-banded (cnt x span_len) matrix A (zero outside each row's interval), B = the staged X
  //      rows (span_len x nx); wave 0 accumulates ceil(span_len/4) v_mfma_f64_16x16x4_f64 per 16 rows x 16 states.
  //      Operand maps (cdna_hip_programming.md §3): A: lane l holds A[l&15][l>>4], B: B[l>>4][l&15],
  //      C/D: col = l&15, row = (l>>4) + 4*reg.  The k-order of the sum differs from the reference's ascending
  //      column loop, so results agree to rounding (~1e-16 relative), not bit for bit. ----
  if constexpr (DXM && WG) {
    if (tid < 64) {
      typedef double d4 __attribute__((ext_vector_type(4)));
      const int lr = tid & 15, kq = tid >> 4;
      const int ksteps = (tl.span_len + 3) >> 2;
      for (int rb = 0; rb < T; rb += 16) {
        const int row = rb + lr;
        const bool row_ok = row < tl.cnt;
        const NodeDev ndr = K.nodes[ph.node0 + tl.k0 + (row_ok ? row : tl.cnt - 1)];
        const int rel0 = ndr.dcol0 - tl.span0;
        const double* drow = Ds + (ndr.drow_off - tl.drow0);
        for (int cb = 0; cb < NX; cb += 16) {
          const int st = cb + lr;
          d4 acc = {0.0, 0.0, 0.0, 0.0};
          for (int s = 0; s < ksteps; ++s) {
            const int kcol = 4 * s + kq;
            const int rel = kcol - rel0;
            const double a = (row_ok && rel >= 0 && rel < ndr.dlen) ? drow[rel] : 0.0;
            const double b = (st < NX && kcol < tl.span_len) ? Xs[st * K.max_span + kcol] : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int orow = rb + kq + 4 * i;
            if (st < NX && orow < tl.cnt) DXs[st * T + orow] = acc[i];
          }
        }
      }
    }
  }

  // ---- perturb this role's variable: h = tol (1+|v|), v+h  (LpFiniteDifferenceDerive.cpp:208-214) ----
  double h = 1.0;
  const int v = role - 1;
  if (WJ && !AN && role >= 1) {
#pragma unroll
    for (int i = 0; i < NX; ++i)
      if (v == i) { h = K.tol * (1 + fabs(xs[i])); xs[i] += h; }
#pragma unroll
    for (int j = 0; j < NU; ++j)
      if (v == NX + j) { h = K.tol * (1 + fabs(us[j])); us[j] += h; }
    if (v == NX + NU) { h = K.tol * (1 + fabs(tk)); tk += h; }
  }
  double f[NX > 0 ? NX : 1], cp[NCs];
#ifdef RPM_DIAG
  if (K.diag_mask & 2) {
    for (int i = 0; i < NX; ++i) f[i] = xs[i] * tk;
    for (int j = 0; j < NCs; ++j) cp[j] = us[0];
  } else
#endif
  if (!AN || role == 0) {
    Prob::dae(ph.phase_num, tk, xs, us, c, f, cp);
  } else if constexpr (AN) {
    Prob::dae_jac_col(ph.phase_num, v, tk, xs, us, c, f, cp);  // f, cp now hold column v of the Jacobian
  }
  if (role == 0 && act) {
#pragma unroll
    for (int i = 0; i < NX; ++i) Fb[i * T + kk] = f[i];
#pragma unroll
    for (int j = 0; j < NC; ++j) {
      Fb[(NX + j) * T + kk] = cp[j];
      if (WG) g[ph.g0 + (NX + j) * ph.N + k] = cp[j];           // path rows, :138-164
    }
  }
  __syncthreads();

  if (act) {
    const int N = ph.N;
    if (WG && sv >= 0 && sv < NX)
      g[ph.g0 + sv * N + k] = (DXM ? DXs[sv * T + kk] : dx) - Fb[sv * T + kk] * (tspan / 2.0);   // defects, :113,122
#ifdef RPM_DIAG
    if (!(K.diag_mask & 8))
#endif
    if (WJ && role >= 1) {
      double J[NO];
#pragma unroll
      for (int o = 0; o < NO; ++o) {
        const double pert = o < NX ? f[o < NX ? o : 0] : cp[o >= NX ? o - NX : 0];
        J[o] = AN ? pert : (pert - Fb[o * T + kk]) / h;
      }
      double* vb = vals + ph.v_nl0 + k;
      if (v < NX + NU) {
        // blocks d/dx_v or d/du_v of every output row (:698-743, :776-796)
#pragma unroll
        for (int o = 0; o < NO; ++o) {
          double val;
          if (o < NX) {
            const double ret = J[o] * (tf - t0) / 2.0;
            val = (o == v) ? ddiag - ret : -ret;          // Ddiag - ret on the diagonal block, :712
          } else {
            val = J[o];
          }
          vb[size_t(o * NB + v) * N] = val;
        }
      } else {
        // d/dt0 and d/dtf blocks (:748-760, :801-811); B-5 sign of the reference kept
        const double a0 = -(tau * 0.5) + 0.5, af = (tau * 0.5) + 0.5;
#pragma unroll
        for (int o = 0; o < NO; ++o) {
          double v0, vf;
          if (o < NX) {
            const double fb = Fb[o * T + kk];
            const double dt = J[o] * (tf - t0) / 2.0;
            v0 = fb * (0.5) - a0 * dt;
            vf = -fb * (0.5) + af * dt;
          } else {
            v0 = a0 * J[o];
            vf = af * J[o];
          }
          vb[size_t(o * NB + NX + NU) * N] = v0;
          vb[size_t(o * NB + NX + NU + 1) * N] = vf;
        }
      }
    }
  }

  // ---- this workgroup's share of the constant Doffdiag block (LpNLPWrapper.cpp:715-718): the block is
  //      nx back-to-back copies of the phase's off-diagonal value list; read each source value once,
  //      store it into every state's copy (all runs contiguous across lanes) ----
#ifdef RPM_DIAG
  if (!(K.diag_mask & 4))
#endif
  if (WJ) {
    const double* __restrict__ src = K.doff_vals + tl.c_src0;
    double* __restrict__ dst = vals + tl.c_dst0;
#pragma unroll
    for (int u = 0; u < CPRE; ++u) {
      const int q = tid + u * nthr;
      if (q < tl.c_cnt) {
#pragma unroll
        for (int i = 0; i < NX; ++i) dst[size_t(i) * tl.c_stride + q] = cpre[u];
      }
    }
    for (int q = tid + CPRE * nthr; q < tl.c_cnt; q += nthr) {
      const double dv = src[q];
#pragma unroll
      for (int i = 0; i < NX; ++i) dst[size_t(i) * tl.c_stride + q] = dv;
    }
  }
}

// ------------------------------------------------------------------------------------------
// rpm_tile_rl_kernel ("role-looped"): the throughput variant for large grids (many instances per launch).
// Same arithmetic and the same output order as rpm_tile_kernel, different thread layout: a workgroup is T nodes x RG
// role GROUPS, and each thread walks the roles g, g+RG, g+2RG, ... of its node one after another.  With T = 64,
// RG = 4 a wave is 64 consecutive nodes of ONE role, so every Jacobian store instruction writes 512 contiguous bytes
// (instead of 4 x 128 B), a launch has 4x fewer workgroups of 4 waves each (one residency round on 256 CUs at 16
// instances of the metric problem), and each workgroup pays its load chain once for 3 dynamics evaluations per thread.
#ifdef RPM_DIAG
#define RPM_TRC(i)                                                                                            \
  if (K.trace && threadIdx.x == 0)                                                                            \
  K.trace[(size_t(blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (i)] = wall_clock64()
#else
#define RPM_TRC(i)
#endif
template <class Prob, int T, int RG, bool WG, bool WJ, bool AN>
__global__ __launch_bounds__(T* RG) void rpm_tile_rl_kernel(const KParams K, const double* __restrict__ xall,
                                                            double* __restrict__ gall, double* __restrict__ vall) {
  constexpr int NX = Prob::NX, NU = Prob::NU, NC = Prob::NC;
  constexpr int NO = NX + NC, NV = NX + NU + 1, NB = NX + NU + 2;
  constexpr int R = WJ ? NV + 1 : (NX > 0 ? NX : 1);
  constexpr int NCs = NC > 0 ? NC : 1;
  constexpr int NTHR = T * RG;
  extern __shared__ double lds[];
  const int tid = threadIdx.x;
  const double* __restrict__ x = xall + size_t(blockIdx.y) * K.n;
  double* __restrict__ g = gall + size_t(blockIdx.y) * K.m;
  double* __restrict__ vals = vall + size_t(blockIdx.y) * K.nnz;
  RPM_TRC(0);
  if (int(blockIdx.x) >= K.n_my_tiles) {
    endpoint_block<Prob, WG, WJ, AN>(K, K.tasks[int(blockIdx.x) - K.n_my_tiles], x, g, vals, lds);
    return;
  }
  const int nt = K.n_my_tiles, per = nt >> 3, rem = nt & 7, xcd = int(blockIdx.x) & 7, slot = int(blockIdx.x) >> 3;
  const TileDev tl = K.tiles[xcd * per + (xcd < rem ? xcd : rem) + slot];
  const TileDev& ph = tl;
  const auto c = (const __attribute__((address_space(4))) double*)K.consts;   // constant address space: scalar loads
  double* Xs = lds;
  double* Us = Xs + NX * K.max_span;
  double* Ds = Us + NU * T;
  double* Fb = Ds + K.max_drow;
  const int kk = tid % T, grp = __builtin_amdgcn_readfirstlane(tid / T);   // a wave is one role group: roles are wave-uniform (scalar branches, scalar block offsets)
  const int kc = kk < tl.cnt ? kk : tl.cnt - 1;
  const int k = tl.k0 + kc;
  const int nidx = ph.node0 + k;
  const double tau = K.points[nidx];
  const NodeDev nd = K.nodes[nidx];
  const double ddiag = WJ ? K.diag[nidx] : 0.0;
#ifdef RPM_DIAG
  const bool diag_noload = K.diag_mask & 2;
#else
  constexpr bool diag_noload = false;
#endif
  for (int q = tid; q < NX * tl.span_len; q += NTHR) {
    const int i = q / tl.span_len, r = q - i * tl.span_len;
    Xs[i * K.max_span + r] = diag_noload ? 1.0e6 + q : x[ph.x_state0 + i * (ph.N + 1) + tl.span0 + r];
  }
  for (int q = tid; q < NU * tl.cnt; q += NTHR) {
    const int j = q / tl.cnt, r = q - j * tl.cnt;
    Us[j * T + r] = diag_noload ? 0.5 : x[ph.x_control0 + j * ph.N + tl.k0 + r];
  }
  if (WG)
    for (int q = tid; q < tl.drow_len; q += NTHR) Ds[q] = diag_noload ? 0.25 : K.dvals[tl.drow0 + q];
  const double t0 = diag_noload ? 0.0 : x[ph.x_t0], tf = diag_noload ? 100.0 : x[ph.x_t0 + 1];
  RPM_TRC(1);
  __syncthreads();
  RPM_TRC(2);

  const bool node_ok = kk < tl.cnt;
  const double tspan = tf - t0;
  const double tk0 = (tau + 1) * (tspan / 2.0) + t0;      // LpNLPWrapper.cpp:80
  const int N = ph.N;
  bool first = true;
  for (int role = grp; role < R || first; role += RG) {
    const bool act = node_ok && role < R;
    double xs[NX > 0 ? NX : 1], us[NU > 0 ? NU : 1];
#pragma unroll
    for (int i = 0; i < NX; ++i) xs[i] = Xs[i * K.max_span + (k - tl.span0)];
#pragma unroll
    for (int j = 0; j < NU; ++j) us[j] = Us[j * T + kc];
    double tk = tk0;
    const int sv = WJ ? role - 1 : role;
    double dx = 0.0;
    if (WG && sv >= 0 && sv < NX) {   // D.X in the reference's ascending-column order (LpSparseMatrix.cpp:142-153)
      const double* drow = Ds + (nd.drow_off - tl.drow0);
      const double* xcol = Xs + sv * K.max_span + (nd.dcol0 - tl.span0);
      for (int j = 0; j < nd.dlen; ++j) dx += drow[j] * xcol[j];
    }
    double h = 1.0;
    const int v = role - 1;
    if (WJ && !AN && role >= 1) {     // h = tol (1+|v|), v+h  (LpFiniteDifferenceDerive.cpp:208-214)
#pragma unroll
      for (int i = 0; i < NX; ++i)
        if (v == i) { h = K.tol * (1 + fabs(xs[i])); xs[i] += h; }
#pragma unroll
      for (int j = 0; j < NU; ++j)
        if (v == NX + j) { h = K.tol * (1 + fabs(us[j])); us[j] += h; }
      if (v == NX + NU) { h = K.tol * (1 + fabs(tk)); tk += h; }
    }
    double f[NX > 0 ? NX : 1], cp[NCs];
#ifdef RPM_DIAG
    if ((K.diag_mask & 1) && role >= 4) {
#pragma unroll
      for (int i = 0; i < NX; ++i) f[i] = Fb[i * T + kk] + h;
#pragma unroll
      for (int j = 0; j < NC; ++j) cp[j] = Fb[(NX + j) * T + kk] + h;
    } else
#endif
    if (!AN || role == 0) {
      Prob::dae(ph.phase_num, tk, xs, us, c, f, cp);
    } else if constexpr (AN) {
      Prob::dae_jac_col(ph.phase_num, v, tk, xs, us, c, f, cp);
    }
    if (first) {   // wave-uniform: the first pass publishes the unperturbed outputs before anyone forms a difference
      if (role == 0 && act) {
#pragma unroll
        for (int i = 0; i < NX; ++i) Fb[i * T + kk] = f[i];
#pragma unroll
        for (int j = 0; j < NC; ++j) {
          Fb[(NX + j) * T + kk] = cp[j];
          if (WG) g[ph.g0 + (NX + j) * N + k] = cp[j];           // path rows, :138-164
        }
      }
      __syncthreads();
      RPM_TRC(3);
      first = false;
    }
    if (act) {
      if (WG && sv >= 0 && sv < NX) g[ph.g0 + sv * N + k] = dx - Fb[sv * T + kk] * (tspan / 2.0);   // defects, :113,122
#ifdef RPM_DIAG
      if (WJ && role >= 1 && !(K.diag_mask & 8)) {
#else
      if (WJ && role >= 1) {
#endif
        double J[NO];
#pragma unroll
        for (int o = 0; o < NO; ++o) {
          const double pert = o < NX ? f[o < NX ? o : 0] : cp[o >= NX ? o - NX : 0];
          J[o] = AN ? pert : (pert - Fb[o * T + kk]) / h;
        }
        double* vb = vals + ph.v_nl0 + k;
        if (v < NX + NU) {            // blocks d/dx_v or d/du_v of every output row (:698-743, :776-796)
#pragma unroll
          for (int o = 0; o < NO; ++o) {
            double val;
            if (o < NX) {
              const double ret = J[o] * (tf - t0) / 2.0;
              val = (o == v) ? ddiag - ret : -ret;
            } else {
              val = J[o];
            }
            vb[size_t(o * NB + v) * N] = val;
          }
        } else {                       // d/dt0 and d/dtf blocks (:748-760, :801-811); B-5 sign kept
          const double a0 = -(tau * 0.5) + 0.5, af = (tau * 0.5) + 0.5;
#pragma unroll
          for (int o = 0; o < NO; ++o) {
            double v0, vf;
            if (o < NX) {
              const double fb = Fb[o * T + kk];
              const double dt = J[o] * (tf - t0) / 2.0;
              v0 = fb * (0.5) - a0 * dt;
              vf = -fb * (0.5) + af * dt;
            } else {
              v0 = a0 * J[o];
              vf = af * J[o];
            }
            vb[size_t(o * NB + NX + NU) * N] = v0;
            vb[size_t(o * NB + NX + NU + 1) * N] = vf;
          }
        }
      }
    }
  }
  RPM_TRC(4);
#ifdef RPM_DIAG
  if (WJ && !(K.diag_mask & 4)) {
#else
  if (WJ) {
#endif
    // this tile's share of the constant Doffdiag block (LpNLPWrapper.cpp:715-718)
    const double* __restrict__ src = K.doff_vals + tl.c_src0;
    double* __restrict__ dst = vals + tl.c_dst0;
    for (int q = tid; q < tl.c_cnt; q += NTHR) {
      const double dv = src[q];
#pragma unroll
      for (int i = 0; i < NX; ++i) dst[size_t(i) * tl.c_stride + q] = dv;
    }
  }
#ifdef RPM_DIAG
  RPM_TRC(5);
  if (K.trace) {
    __builtin_amdgcn_s_waitcnt(0);
    RPM_TRC(6);
    if (threadIdx.x == 0) {
      unsigned hw;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      K.trace[(size_t(blockIdx.y) * gridDim.x + blockIdx.x) * 8 + 7] = (static_cast<unsigned long long>(xcc) << 32) | hw;
    }
  }
#endif
}

// ------------------------------------------------------------------------------------------
// rpm_tile_pl_kernel ("pipelined"): the role-looped layout made persistent and wave-specialised.  Why: the per-
// workgroup timeline of rpm_tile_rl_kernel is serial (input loads 4-5 us behind the store traffic, 3 dynamics passes,
// then a 2.5 us burst of constant-block stores that blocks the issuing waves), every workgroup of a launch is in the
// same phase at the same time, and a launch is only two residency rounds, so neither the SIMDs (busy 35 %) nor HBM
// (busy 45 %) are kept fed (tools/trace_timeline.py).  Here a workgroup is NH independent halves of RG compute waves +
// NDMA DMA waves (pl_shape), and a half walks tiles w, w+G, w+2G, ...:
//   * the DMA waves copy the NEXT tile's inputs (tile record, t0 tf, X rows, U rows, D rows, node records, its slice
//     of the constant list) from HBM straight into the other LDS staging buffer (global_load_lds_dwordx4) while the
//     compute waves work on the current one, and write the current tile's share of the constant Doffdiag block, so
//     the compute waves never wait for a load or a store burst; they run at raised priority (s_setprio);
//   * the compute waves run exactly the role loop of rpm_tile_rl_kernel (same arithmetic, same output order:
//     bit-identical results, tests/test_gpu_parity.py) out of the staged buffer;
//   * endpoint work items (events, linkages, linear rows) are taken by DMA waves once their tiles are done, one wave
//     each (endpoint_block<..., WAVE = true>).
// One workgroup barrier per tile (A: staging buffer ready; the DMA waves execute s_waitcnt vmcnt(0) before it); F
// (unperturbed dynamics published) is a flag in LDS that only the compute waves look at, so the DMA waves spend the
// first pass — when the compute waves store nothing — on the constant block.  Host-checked limits: a tile's constant
// share <= PL_CMAX doubles (it passes through registers of the DMA waves), 2 NX + 3 <= 64 (endpoint perturbations
// fit one wave).
constexpr int PL_CMAX = 1280, PL_REC = 32;   // a tile's constant share: at most PL_CMAX doubles
typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));   // a pair of doubles at 8-byte alignment
#define RPM_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define RPM_LPTR(p) ((__attribute__((address_space(3))) void*)(p))
// One wave copies `len` doubles from global memory straight into LDS (both sides 8-byte aligned).  Inlined (a call
// would start with s_waitcnt 0 and serialise the loads) but not unrolled: the DMA wave runs this code once per tile,
// so it should be small enough to stay in the instruction cache.
template <int NPART>
__device__ __forceinline__ void pl_dma_run(const double* gsrc, double* ldst, int len, int lane, int part) {
  const int pairs = len >> 1;   // chunk ch (64 pairs) is copied by the wave with part == ch % NPART
#pragma unroll 1
  for (int ch = part; ch * 64 < pairs; ch += NPART)
    if (ch * 64 + lane < pairs)
      __builtin_amdgcn_global_load_lds(RPM_GPTR(gsrc + ch * 128 + 2 * lane), RPM_LPTR(ldst + ch * 128), 16, 0, 0);
  if ((len & 1) && part == 0 && lane < 2)   // odd tail: the last double as two dwords
    __builtin_amdgcn_global_load_lds(RPM_GPTR(reinterpret_cast<const int*>(gsrc + len - 1) + lane),
                                     RPM_LPTR(ldst + len - 1), 4, 0, 0);
}


// role groups (= compute waves) of the pipelined kernel for a problem with R = nx + nu + 2 roles: one role per wave
// when they fit (R <= 12; with the 2 DMA waves 14 waves = 4 per SIMD, which the launch bound turns into a 128-VGPR
// budget), else the fewest equal passes (R = 18: two passes of 9 waves, 11 waves = 3 per SIMD, 168 VGPRs)
constexpr int pl_role_groups(int R) { return (R + (R + 11) / 12 - 1) / ((R + 11) / 12); }

// Shape of the pipelined kernel's workgroup for a problem with R = nx + nu + 2 roles: NH independent halves, each RG
// compute waves (roles g, g + RG, ...) + NDMA DMA waves.  R <= 12: two halves of 4 + 2 waves = 12 waves, 3 per SIMD
// (2 compute + 1 DMA, 168-VGPR budget), three passes per tile — measured best on the metric problem (33.2 us per
// 16-iterate launch) against 2 x (6 + 2) at 128 VGPRs (35.0), one role per wave 12 + 4 (36.3) and 3 x (4 + 1) (38.6).
// Larger problems: one half, the fewest equal passes (R = 18: 9 + 2 waves, 3 per SIMD).
struct PlShape { int NH, RG, NDMA; };
constexpr PlShape pl_shape(int R) {
  return R <= 12 ? PlShape{2, 4, 2} : PlShape{1, pl_role_groups(R), 2};
}

template <class Prob, int NH, int RG, int NDMA, bool WG, bool WJ, bool AN>
__global__ __launch_bounds__(NH * 64 * (RG + NDMA)) void rpm_tile_pl_kernel(
    const KParams K, int n_inst, const double* __restrict__ xall, double* __restrict__ gall,
    double* __restrict__ vall) {
  constexpr int T = 64;   // a role of a tile is one wave
  constexpr int HT = 64 * (RG + NDMA);   // threads of one half
  constexpr int CCH = (PL_CMAX / 128 + NDMA - 1) / NDMA;   // 128-double chunks of the constant share per DMA wave
  constexpr int NX = Prob::NX, NU = Prob::NU, NC = Prob::NC;
  constexpr int NXs = NX > 0 ? NX : 1, NUs = NU > 0 ? NU : 1;
  constexpr int NO = NX + NC, NV = NX + NU + 1, NB = NX + NU + 2;
  constexpr int R = WJ ? NV + 1 : (NX > 0 ? NX : 1);
  constexpr int NCs = NC > 0 ? NC : 1;
  constexpr int NTHR = T * RG;
  constexpr int NREC = int(sizeof(TileDev) / sizeof(int));
  static_assert(NREC < PL_REC, "tile record plus the instance index must fit the staged record");
  extern __shared__ double lds_all[];
  // the halves of a workgroup are independent (own tiles, own LDS); they only share the barriers
  const int half = NH > 1 ? __builtin_amdgcn_readfirstlane(int(threadIdx.x) / HT) : 0;   // wave-uniform
  const int tid = int(threadIdx.x) - half * HT;
  const int G = NH * int(gridDim.x), w = NH * int(blockIdx.x) + half;
  const int nt = K.n_my_tiles;
  const int W = nt * n_inst;
  const int n_iter = w < W ? (W - w + G - 1) / G : 0;                 // tiles w, w + G, ... of this half
  const int n_iter_wg = (W - NH * int(blockIdx.x) + G - 1) / G;       // of half 0: the barrier count of the workgroup
  // one staging buffer (doubles): record, next tile's record | t0 tf | X rows | U rows | D rows | tau | diag | node
  // records | const share
  const int S_TT = PL_REC, S_X = S_TT + 2, S_U = S_X + NX * K.max_span, S_D = S_U + NU * T;
  const int S_TAU = S_D + K.max_drow, S_DG = S_TAU + T, S_ND = S_DG + T, S_CV = S_ND + 2 * T;
  const int S_SIZE = S_CV + (WJ ? K.max_cshare : 0);
  double* lds = lds_all + half * (2 * S_SIZE + (NX + NC) * T + 2);
  double* Fb = lds + 2 * S_SIZE;
  int* fb_ready = reinterpret_cast<int*>(Fb + (NX + NC) * T);   // tile count for which Fb holds the unperturbed dynamics
#ifdef RPM_DIAG
#define RPM_PTRC(j, slot)                                                           \
  if (K.trace && (threadIdx.x & 63) == 0 && (j) < 2) K.trace[size_t(w) * 64 + (j)*32 + (slot)] = wall_clock64()
#else
#define RPM_PTRC(j, slot)
#endif
  if (tid == 0) { RPM_PTRC(0, 31); }

  if (tid >= NTHR) {
    // ---------------- DMA waves (two: a direct-to-LDS load takes ~60 ns to issue, so the runs of a tile and the
    // chunks of the constant block are dealt alternately to them) ----------------
    const int lane = (tid - NTHR) & 63;
    const int dw = __builtin_amdgcn_readfirstlane((tid - NTHR) >> 6);
    // The DMA waves' instruction stream is long and scalar; sharing a SIMD with three busy compute waves it would get
    // a quarter of the issue slots (4 us to issue one tile's loads).  They run at raised priority instead.
    __builtin_amdgcn_s_setprio(3);
    // The next tile's inputs go from global memory straight into the other LDS staging buffer
    // (global_load_lds_dwordx4, 16 B per lane, no VGPR round trip): the DMA wave only issues them.
    // every run is dealt chunk-wise to the NDMA waves, the first chunk of successive runs to successive waves
    int rot = 0;
    auto run = [&](const double* gsrc, double* ldst, int len) {
      pl_dma_run<NDMA>(gsrc, ldst, len, lane, (dw + NDMA - (rot++ & (NDMA - 1))) & (NDMA - 1));
    };
    // The addresses of a tile's runs come from its record.  In steady state that record is already in LDS (each
    // staging buffer also carries the record of the tile AFTER its own), so issuing the next tile's loads never waits
    // for global memory; only the first tile of a workgroup reads its record from HBM.
    struct TileRuns { int k0, cnt, span0, span_len, drow0, drow_len, N, x_state0, x_control0, x_t0, node0, c_src0, c_cnt; };
    auto runs_of = [&](const int* p) {
      TileRuns r;
#define RPM_RF(f) r.f = __builtin_amdgcn_readfirstlane(p[offsetof(TileDev, f) / 4])
      RPM_RF(k0); RPM_RF(cnt); RPM_RF(span0); RPM_RF(span_len); RPM_RF(drow0); RPM_RF(drow_len); RPM_RF(N);
      RPM_RF(x_state0); RPM_RF(x_control0); RPM_RF(x_t0); RPM_RF(node0); RPM_RF(c_src0); RPM_RF(c_cnt);
#undef RPM_RF
      return r;
    };
    auto stage = [&](int item, double* buf, const TileRuns tl) {
      const int inst = item / nt, tidx = item - inst * nt;
      const double* __restrict__ x = xall + size_t(inst) * K.n;
      static_assert(NREC % 2 == 0 && sizeof(TileDev) % 8 == 0, "the tile record is copied as doubles");
      rot = 0;
      if (dw == 0 && lane == 0) reinterpret_cast<int*>(buf)[NREC] = inst;   // before the direct loads: an LDS write after them waits for them
      run(reinterpret_cast<const double*>(K.tiles + tidx), buf, NREC / 2);
      if (item + G < W) {   // the record of this workgroup's tile after this one
        const int item2 = item + G, inst2 = item2 / nt;
        run(reinterpret_cast<const double*>(K.tiles + (item2 - inst2 * nt)), buf + PL_REC / 2, NREC / 2);
      }
      run(x + tl.x_t0, buf + S_TT, 2);
#pragma unroll
      for (int i = 0; i < NX; ++i) run(x + tl.x_state0 + i * (tl.N + 1) + tl.span0, buf + S_X + i * K.max_span, tl.span_len);
#pragma unroll
      for (int j = 0; j < NU; ++j) run(x + tl.x_control0 + j * tl.N + tl.k0, buf + S_U + j * T, tl.cnt);
      run(K.points + tl.node0 + tl.k0, buf + S_TAU, tl.cnt);
      if (WJ) run(K.diag + tl.node0 + tl.k0, buf + S_DG, tl.cnt);
      run(reinterpret_cast<const double*>(K.nodes + tl.node0 + tl.k0), buf + S_ND, 2 * tl.cnt);
      if (WG) run(K.dvals + tl.drow0, buf + S_D, tl.drow_len);
      if (WJ) run(K.doff_vals + tl.c_src0, buf + S_CV, tl.c_cnt);
    };
    if (dw == 0 && lane == 0) *fb_ready = 0;
    if (n_iter > 0) stage(w, lds, runs_of(reinterpret_cast<const int*>(K.tiles + (w - (w / nt) * nt))));
    for (int j = 0; j < n_iter_wg; ++j) {
      const double* cur = lds + (j & 1) * S_SIZE;
      double* nxt = lds + ((j + 1) & 1) * S_SIZE;
      __builtin_amdgcn_s_waitcnt(0);   // the staged loads (and the constant stores before them) have landed
      __syncthreads();                 // A: buffer `cur` is complete
      RPM_PTRC(j, 16);
      // this tile's share of the constant Doffdiag block (LpNLPWrapper.cpp:715-718), written while the compute waves are
      // in their first pass and store nothing; then the next tile's loads.  (The order matters twice: an LDS read of
      // this wave after the direct-to-LDS loads would wait for them, and the Jacobian stores of the later passes
      // should not meet these in the memory system.)
      if (WJ && j < n_iter) {
        const int* rec = reinterpret_cast<const int*>(cur);
        const int c_dst0 = __builtin_amdgcn_readfirstlane(rec[offsetof(TileDev, c_dst0) / 4]);
        const int inst = __builtin_amdgcn_readfirstlane(rec[NREC]);
        const int c_cnt = __builtin_amdgcn_readfirstlane(rec[offsetof(TileDev, c_cnt) / 4]);
        const int c_stride = __builtin_amdgcn_readfirstlane(rec[offsetof(TileDev, c_stride) / 4]);
        double* __restrict__ cdst = vall + size_t(inst) * K.nnz + c_dst0;
        d2u cv[CCH];
#pragma unroll
        for (int ch = 0; ch < CCH; ++ch) {
          const int q = min((NDMA * ch + dw) * 128 + 2 * lane, c_cnt - 2);
          cv[ch].x = cur[S_CV + q];
          cv[ch].y = cur[S_CV + q + 1];
        }
        const double ctail = cur[S_CV + c_cnt - 1];
#pragma unroll
        for (int ch = 0; ch < CCH; ++ch) {
          const int q = (NDMA * ch + dw) * 128 + 2 * lane;   // 16 B per lane: 1 KB per store instruction
          if (q + 1 < c_cnt) {
#pragma unroll
            for (int i = 0; i < NX; ++i) *reinterpret_cast<d2u*>(cdst + size_t(i) * c_stride + q) = cv[ch];
          }
        }
        if ((c_cnt & 1) && dw == 0 && lane < NX) cdst[size_t(lane) * c_stride + c_cnt - 1] = ctail;
      }
      RPM_PTRC(j, 17);
      if (j + 1 < n_iter) stage(w + (j + 1) * G, nxt, runs_of(reinterpret_cast<const int*>(cur) + PL_REC));
      RPM_PTRC(j, 18);
    }
    // endpoint work items of this workgroup, one wave each
    const int n_end = K.n_tasks * n_inst;
    for (int it = NDMA * w + dw; it < n_end; it += NDMA * G) {
      const int inst = it / K.n_tasks;
      endpoint_block<Prob, WG, WJ, AN, true>(K, K.tasks[it - inst * K.n_tasks], xall + size_t(inst) * K.n,
                                             gall + size_t(inst) * K.m, vall + size_t(inst) * K.nnz, nullptr);
    }
    return;
  }

#ifdef RPM_DIAG
#define RPM_JSTORE(dst, val) if (!(K.diag_mask & 8) || (val) == 1e300) dst = (val)
#else
#define RPM_JSTORE(dst, val) dst = (val)
#endif
  // ---------------- compute waves: the role loop of rpm_tile_rl_kernel out of the staged buffer ----------------
  // the problem constants through the constant address space: scalar loads (s_load, lgkmcnt).  Through a generic
  // pointer they are vector loads here (the kernel has stored by then, so the compiler cannot use the scalar cache),
  // and a vector load's s_waitcnt vmcnt also waits for every Jacobian store issued before it (in-order counter).
  const auto c4 = (const __attribute__((address_space(4))) double*)K.consts;
  const int kk = tid % T, grp = __builtin_amdgcn_readfirstlane(tid / T);   // a wave is one role group: roles are wave-uniform (scalar branches, scalar block offsets)
  for (int jt = 0; jt < n_iter_wg; ++jt) {
    const double* cur = lds + (jt & 1) * S_SIZE;
    __syncthreads();   // A
    if (jt >= n_iter) continue;   // the other half still has a tile: keep the barrier count
    if (grp < 4) { RPM_PTRC(jt, grp * 4 + 0); }
    const int* rec = reinterpret_cast<const int*>(cur);
    const int k0 = __builtin_amdgcn_readfirstlane(rec[offsetof(TileDev, k0) / 4]);
    const int cnt = __builtin_amdgcn_readfirstlane(rec[offsetof(TileDev, cnt) / 4]);
    const int span0 = __builtin_amdgcn_readfirstlane(rec[offsetof(TileDev, span0) / 4]);
    const int drow0 = __builtin_amdgcn_readfirstlane(rec[offsetof(TileDev, drow0) / 4]);
    const int N = __builtin_amdgcn_readfirstlane(rec[offsetof(TileDev, N) / 4]);
    const int phase_num = __builtin_amdgcn_readfirstlane(rec[offsetof(TileDev, phase_num) / 4]);
    const int g0 = __builtin_amdgcn_readfirstlane(rec[offsetof(TileDev, g0) / 4]);
    const int v_nl0 = __builtin_amdgcn_readfirstlane(rec[offsetof(TileDev, v_nl0) / 4]);
    const int inst = __builtin_amdgcn_readfirstlane(rec[NREC]);
    double* __restrict__ g = gall + size_t(inst) * K.m;
    double* __restrict__ vals = vall + size_t(inst) * K.nnz;
    const double* Xs = cur + S_X;
    const double* Us = cur + S_U;
    const double* Ds = cur + S_D;
    const int kc = kk < cnt ? kk : cnt - 1;
    const int k = k0 + kc;
    const bool node_ok = kk < cnt;
    bool first = true;
    // per-pass scalars (tau, t0, tf, the node record, the diagonal of D) are re-read from LDS where they are used
    // instead of living in registers across the dynamics call: the 10-wave workgroup has 168 VGPRs per lane
    for (int role = grp; role < R || first; role += RG) {
      const bool act = node_ok && role < R;
#ifdef RPM_DIAG
      const bool trc = role == 5;
      if (trc) { RPM_PTRC(jt, 24); }
#endif
      double xs[NXs], us[NUs];
#pragma unroll
      for (int i = 0; i < NX; ++i) xs[i] = Xs[i * K.max_span + (k - span0)];
#pragma unroll
      for (int j = 0; j < NU; ++j) us[j] = Us[j * T + kc];
      double tk;
      {
        const double tau = cur[S_TAU + kc], t0 = cur[S_TT], tf = cur[S_TT + 1];
        tk = (tau + 1) * ((tf - t0) / 2.0) + t0;      // LpNLPWrapper.cpp:80
      }
      const int sv = WJ ? role - 1 : role;
      double dx = 0.0;
      if (WG && sv >= 0 && sv < NX) {   // D.X in the reference's ascending-column order (LpSparseMatrix.cpp:142-153)
        const NodeDev nd = reinterpret_cast<const NodeDev*>(cur + S_ND)[kc];
        const double* drow = Ds + (nd.drow_off - drow0);
        const double* xcol = Xs + sv * K.max_span + (nd.dcol0 - span0);
        // same ascending order, operands fetched four columns at a time so that the LDS latency is paid per batch
        int j = 0;
        for (; j + 4 <= nd.dlen; j += 4) {
          const double d0 = drow[j], d1 = drow[j + 1], d2 = drow[j + 2], d3 = drow[j + 3];
          const double x0 = xcol[j], x1 = xcol[j + 1], x2 = xcol[j + 2], x3 = xcol[j + 3];
          dx += d0 * x0;
          dx += d1 * x1;
          dx += d2 * x2;
          dx += d3 * x3;
        }
        for (; j < nd.dlen; ++j) dx += drow[j] * xcol[j];
      }
#ifdef RPM_DIAG
      if (trc) { if (dx == 1e300) xs[0] = 0; RPM_PTRC(jt, 25); }
#endif
      double h = 1.0;
      const int v = role - 1;
      if (WJ && !AN && role >= 1) {     // h = tol (1+|v|), v+h  (LpFiniteDifferenceDerive.cpp:208-214)
        // the role is wave-uniform: fetch the one perturbed variable by its (scalar) row, form h and v+h once, and
        // put the sum back where it belongs — instead of forming them for every variable and selecting
        double pv;
        if (v < NX) pv = Xs[v * K.max_span + (k - span0)];
        else if (v < NX + NU) pv = Us[(v - NX) * T + kc];
        else pv = tk;
        h = K.tol * (1 + fabs(pv));
        const double pp = pv + h;
#pragma unroll
        for (int i = 0; i < NX; ++i) xs[i] = (v == i) ? pp : xs[i];
#pragma unroll
        for (int j = 0; j < NU; ++j) us[j] = (v == NX + j) ? pp : us[j];
        tk = (v == NX + NU) ? pp : tk;
      }
      double f[NXs], cp[NCs];
      if (!AN || role == 0) {
        Prob::dae(phase_num, tk, xs, us, c4, f, cp);
      } else if constexpr (AN) {
        Prob::dae_jac_col(phase_num, v, tk, xs, us, c4, f, cp);
      }
#ifdef RPM_DIAG
      if (trc) { if (f[0] == 1e300) cp[0] = 0; RPM_PTRC(jt, 26); }
#endif
      if (first) {   // wave-uniform: the first pass publishes the unperturbed outputs before anyone forms a difference
        if (role == 0 && act) {
#pragma unroll
          for (int i = 0; i < NX; ++i) Fb[i * T + kk] = f[i];
#pragma unroll
          for (int j = 0; j < NC; ++j) {
            Fb[(NX + j) * T + kk] = cp[j];
            if (WG) g[g0 + (NX + j) * N + k] = cp[j];           // path rows, :138-164
          }
        }
        if (grp < 4) { RPM_PTRC(jt, grp * 4 + 1); }
        // F: the other compute waves wait for role 0's outputs.  A flag in LDS, not s_barrier: the DMA waves stay out
        // of it (they are busy with the constant block and the next tile), and the role-0 wave never waits.
        if (grp == 0) {
          __hip_atomic_store(fb_ready, jt + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
          while (__hip_atomic_load(fb_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < jt + 1)
            __builtin_amdgcn_s_sleep(1);
        }
        if (grp < 4) { RPM_PTRC(jt, grp * 4 + 2); }
        first = false;
      }
      if (act) {
        const double tau = cur[S_TAU + kc], t0 = cur[S_TT], tf = cur[S_TT + 1];
        const double ddiag = cur[S_DG + kc];
        if (WG && sv >= 0 && sv < NX) g[g0 + sv * N + k] = dx - Fb[sv * T + kk] * ((tf - t0) / 2.0);   // defects, :113,122
        if (WJ && role >= 1) {
          double J[NO];
#pragma unroll
          for (int o = 0; o < NO; ++o) {
            const double pert = o < NX ? f[o < NX ? o : 0] : cp[o >= NX ? o - NX : 0];
            J[o] = AN ? pert : (pert - Fb[o * T + kk]) / h;
          }
          double* __restrict__ vb = vals + v_nl0;   // block bases stay scalar; the node index k is the only per-lane part
          if (v < NX + NU) {            // blocks d/dx_v or d/du_v of every output row (:698-743, :776-796)
#pragma unroll
            for (int o = 0; o < NO; ++o) {
              double val;
              if (o < NX) {
                const double ret = J[o] * (tf - t0) / 2.0;
                val = (o == v) ? ddiag - ret : -ret;
              } else {
                val = J[o];
              }
              RPM_JSTORE((vb + size_t(o * NB + v) * N)[k], val);
            }
          } else {                       // d/dt0 and d/dtf blocks (:748-760, :801-811); B-5 sign kept
            const double a0 = -(tau * 0.5) + 0.5, af = (tau * 0.5) + 0.5;
#pragma unroll
            for (int o = 0; o < NO; ++o) {
              double v0, vf;
              if (o < NX) {
                const double fb = Fb[o * T + kk];
                const double dt = J[o] * (tf - t0) / 2.0;
                v0 = fb * (0.5) - a0 * dt;
                vf = -fb * (0.5) + af * dt;
              } else {
                v0 = a0 * J[o];
                vf = af * J[o];
              }
              RPM_JSTORE((vb + size_t(o * NB + NX + NU) * N)[k], v0);
              RPM_JSTORE((vb + size_t(o * NB + NX + NU + 1) * N)[k], vf);
            }
          }
        }
      }
#ifdef RPM_DIAG
      if (trc) { RPM_PTRC(jt, 27); }
#endif
    }
    if (grp < 4) { RPM_PTRC(jt, grp * 4 + 3); }
  }
}

// ------------------------------------------------------------------------------------------
// Objective and gradient.  One workgroup per phase; thread = node (strided).  Sums use a fixed
// binary tree over the workgroup so the result is deterministic (independent of timing).
template <class Prob, bool GRAD, bool AN>
__global__ void rpm_obj_kernel(const KParams K, const double* __restrict__ xall, double* __restrict__ objall,
                               double* __restrict__ gradall, double* __restrict__ partial) {
  constexpr int NX = Prob::NX, NU = Prob::NU;
  constexpr int NXs = NX > 0 ? NX : 1, NUs = NU > 0 ? NU : 1;
  __shared__ double red[3][256];
  const int tid = threadIdx.x;
  const int p = blockIdx.x;
  const int inst = blockIdx.y;
  const double* __restrict__ x = xall + size_t(inst) * K.n;
  double* grad = GRAD ? gradall + size_t(inst) * K.n : nullptr;
  const PhaseDev ph = K.phases[p];
  const double* c = K.consts;
  const int N = ph.N;
  const double t0 = x[ph.x_t0], tf = x[ph.x_t0 + 1];
  const double tspan = tf - t0;
  double s_wl = 0.0, s_t0 = 0.0;   // sum w_k L_k ; sum (w_k dt/2 dL/dt)_k (1-tau_k)/2
  for (int k = tid; k < N; k += blockDim.x) {
    const double tau = K.points[ph.node0 + k], w = K.weights[ph.node0 + k];
    const double tk = (tau + 1) * (tspan / 2.0) + t0;
    double xs[NXs], us[NUs];
#pragma unroll
    for (int i = 0; i < NX; ++i) xs[i] = x[ph.x_state0 + i * (N + 1) + k];
#pragma unroll
    for (int j = 0; j < NU; ++j) us[j] = x[ph.x_control0 + j * N + k];
    const double L0 = Prob::lagrange(ph.phase_num, tk, xs, us, c);
    s_wl += w * L0;
    if (GRAD) {
      const double wk = w * tspan / 2.0;                       // Weights*tspan/2.0, :1051
      double dLt;
      if constexpr (AN) {
#pragma unroll
        for (int i = 0; i < NX; ++i)
          grad[ph.x_state0 + i * (N + 1) + k] = wk * Prob::lagrange_grad_col(ph.phase_num, i, tk, xs, us, c);
#pragma unroll
        for (int j = 0; j < NU; ++j)
          grad[ph.x_control0 + j * N + k] = wk * Prob::lagrange_grad_col(ph.phase_num, NX + j, tk, xs, us, c);
        dLt = Prob::lagrange_grad_col(ph.phase_num, NX + NU, tk, xs, us, c);
      } else {
        // LpFDderive::DerivLagrange, LpFiniteDifferenceDerive.cpp:100-192
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          const double b = xs[i], hh = K.tol * (1 + fabs(b));
          xs[i] = b + hh;
          const double Lp = Prob::lagrange(ph.phase_num, tk, xs, us, c);
          xs[i] = b;
          grad[ph.x_state0 + i * (N + 1) + k] = wk * ((Lp - L0) / hh);
        }
#pragma unroll
        for (int j = 0; j < NU; ++j) {
          const double b = us[j], hh = K.tol * (1 + fabs(b));
          us[j] = b + hh;
          const double Lp = Prob::lagrange(ph.phase_num, tk, xs, us, c);
          us[j] = b;
          grad[ph.x_control0 + j * N + k] = wk * ((Lp - L0) / hh);
        }
        const double ht = K.tol * (1 + fabs(tk));
        dLt = (Prob::lagrange(ph.phase_num, tk + ht, xs, us, c) - L0) / ht;
      }
      s_t0 += ((w * (tspan / 2.0)) * dLt) * (tau * (-0.5) + 0.5);   // ret2*ret3, :1072-1077
    }
  }
  red[0][tid] = s_wl;
  red[1][tid] = s_t0;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if (tid < s) {
      red[0][tid] += red[0][tid + s];
      red[1][tid] += red[1][tid + s];
    }
    __syncthreads();
  }
  if (tid == 0) {
    double x0[NXs], xf[NXs];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      x0[i] = x[ph.x_state0 + i * (N + 1)];
      xf[i] = x[ph.x_state0 + i * (N + 1) + N];
    }
    const double wl = red[0][0];
    const double mayer = Prob::mayer(ph.phase_num, t0, x0, tf, xf, c);
    // per-phase cost  Mayer + (w'L)(dt/2)  (:926-932); phases are summed in order by phase 0's thread below
    partial[size_t(inst) * K.P + p] = mayer + wl * (tspan / 2.0);
    if (GRAD) {
      // Mayer derivative w.r.t. [x0.., t0, xf.., tf]  (LpFDderive::DerivMayer :11-98 or the analytic callback)
      double dM[2 * NXs + 2];
      if constexpr (AN) {
        for (int q = 0; q < 2 * NX + 2; ++q) dM[q] = Prob::mayer_grad_col(ph.phase_num, q, t0, x0, tf, xf, c);
      } else {
        const double m0 = mayer;
        const double h0 = K.tol * (1 + fabs(t0)), hf = K.tol * (1 + fabs(tf));
        dM[NX] = (Prob::mayer(ph.phase_num, t0 + h0, x0, tf, xf, c) - m0) / h0;
        dM[2 * NX + 1] = (Prob::mayer(ph.phase_num, t0, x0, tf + hf, xf, c) - m0) / hf;
        for (int i = 0; i < NX; ++i) {
          const double b0 = x0[i], hb0 = K.tol * (1 + fabs(b0));
          x0[i] = b0 + hb0;
          dM[i] = (Prob::mayer(ph.phase_num, t0, x0, tf, xf, c) - m0) / hb0;
          x0[i] = b0;
          const double bf = xf[i], hbf = K.tol * (1 + fabs(bf));
          xf[i] = bf + hbf;
          dM[NX + 1 + i] = (Prob::mayer(ph.phase_num, t0, x0, tf, xf, c) - m0) / hbf;
          xf[i] = bf;
        }
      }
      // terminal-state entries (:1054).  The initial-state Mayer entry is overwritten by the Lagrange
      // run in the reference (:1050-1053) — kept, it is zero in every supported problem anyway.
      for (int i = 0; i < NX; ++i) grad[ph.x_state0 + i * (N + 1) + N] = dM[NX + 1 + i];
      // d/dt0 (:1069-1078) and d/dtf (:1081-1087, which keeps only node 0 of the dL/dt term)
      grad[ph.x_t0] = (red[1][0] + dM[NX]) + (-0.5) * wl;
      double dLt0;
      {
        double xs[NXs], us[NUs];
        for (int i = 0; i < NX; ++i) xs[i] = x0[i];
        for (int j = 0; j < NU; ++j) us[j] = x[ph.x_control0 + j * N];
        const double tau = K.points[ph.node0];
        const double tk = (tau + 1) * (tspan / 2.0) + t0;
        if constexpr (AN) {
          dLt0 = Prob::lagrange_grad_col(ph.phase_num, NX + NU, tk, xs, us, c);
        } else {
          const double ht = K.tol * (1 + fabs(tk));
          dLt0 = (Prob::lagrange(ph.phase_num, tk + ht, xs, us, c) - Prob::lagrange(ph.phase_num, tk, xs, us, c)) / ht;
        }
        const double r2 = (K.weights[ph.node0] * (tspan / 2.0)) * dLt0;
        grad[ph.x_t0 + 1] = (dM[2 * NX + 1] + 0.5 * wl) + (tau * 0.5 + 0.5) * r2;
      }
    }
  }
  (void)objall;
}

// sum the per-phase costs in phase order (GetObjFun's `cost +=` loop, :872-937)
__global__ void rpm_obj_sum_kernel(int P, int B, const double* __restrict__ partial, double* __restrict__ obj) {
  const int inst = blockIdx.x * blockDim.x + threadIdx.x;
  if (inst >= B) return;
  double cost = 0.0;
  for (int p = 0; p < P; ++p) cost += partial[size_t(inst) * P + p];
  obj[inst] = cost;
}

// ------------------------------------------------------------------------------------------
// Exact-Hessian mode (hessian-approximation=exact): forward SECOND differences of the user functions
// (LpHessianCalculator::CalculatePhaseHessian, Core/LpHessian.cpp:1192-2161), lambda-weighted and assembled as in
// GetPhaseHessian (:12-599).  Per node there are NR = (NV+1)(NV+2)/2 evaluation points (base, NV single and
// NV(NV+1)/2 double perturbations of [x.., u.., t]); thread = (role, node): every point is evaluated concurrently,
// published in LDS, then each pair role combines F_ab - F_a - F_b + F_0 and writes its N-long block.

template <class Prob>
__global__ void rpm_dep_probe_kernel(const KParams K, const double* __restrict__ xg, int* __restrict__ dep,
                                     const int* __restrict__ dep_off) {
  // NaN-propagation probe at node 1 of the guess (LpDerivDependciesChecker.cpp:60-93)
  constexpr int NX = Prob::NX, NU = Prob::NU, NC = Prob::NC;
  const PhaseDev ph = K.phases[blockIdx.x];
  const int v = threadIdx.x;
  if (v >= NX + NU) return;
  const double t0 = xg[ph.x_t0], tf = xg[ph.x_t0 + 1];
  const double tk = (K.points[ph.node0 + 1] + 1) * ((tf - t0) / 2.0) + t0;
  double xs[NX > 0 ? NX : 1], us[NU > 0 ? NU : 1], f[NX > 0 ? NX : 1], cp[NC > 0 ? NC : 1];
  for (int i = 0; i < NX; ++i) xs[i] = (i == v) ? __builtin_nan("") : xg[ph.x_state0 + i * (ph.N + 1) + 1];
  for (int j = 0; j < NU; ++j) us[j] = (NX + j == v) ? __builtin_nan("") : xg[ph.x_control0 + j * ph.N + 1];
  Prob::dae(ph.phase_num, tk, xs, us, K.consts, f, cp);
  int* out = dep + dep_off[blockIdx.x] + v * (NX + NC);
  for (int r = 0; r < NX; ++r) out[r] = isfinite(f[r]) ? 0 : 1;
  for (int r = 0; r < NC; ++r) out[NX + r] = isfinite(cp[r]) ? 0 : 1;
}

template <class Prob, bool AN>
__global__ void rpm_hess_kernel(const KParams K, const HParams Hp, const double* __restrict__ xall, const double sigma,
                                const double* __restrict__ lam_all, double* __restrict__ hv_all,
                                double* __restrict__ tmp_all) {
  constexpr int NX = Prob::NX, NU = Prob::NU, NC = Prob::NC;
  constexpr int NV = NX + NU + 1, NF = NX + NC + 1, NR = (NV + 1) * (NV + 2) / 2;
  constexpr int NXs = NX > 0 ? NX : 1, NUs = NU > 0 ? NU : 1, NCs = NC > 0 ? NC : 1;
  extern __shared__ double lds[];   // F values: [(role*NF + o)*TH + node]
  const int TH = Hp.th;
  const int tid = threadIdx.x;
  const int kk = tid % TH, role = tid / TH;
  const int* tile = Hp.tiles + 3 * blockIdx.x;
  const PhaseDev ph = K.phases[tile[0]];
  const HessPhaseDev hp = Hp.phases[tile[0]];
  const int k0 = tile[1], cnt = tile[2];
  const int inst = blockIdx.y;
  const double* __restrict__ x = xall + size_t(inst) * K.n;
  const double* __restrict__ lam = lam_all + size_t(inst) * K.m + ph.g0;   // phase_lambda, LpHessian.cpp:84
  double* __restrict__ hv = hv_all + size_t(inst) * Hp.nnz_h + hp.v0;
  double* __restrict__ tmp = tmp_all + size_t(inst) * Hp.tmp_len + hp.tt_tmp;
  const double* c = K.consts;
  const bool act = role < NR && kk < cnt;
  const int k = k0 + (kk < cnt ? kk : cnt - 1);
  const int N = ph.N;
  const double t0 = x[ph.x_t0], tf = x[ph.x_t0 + 1];
  const double tau = K.points[ph.node0 + k], wq = K.weights[ph.node0 + k];
  const double tk0 = (tau + 1) * ((tf - t0) / 2.0) + t0;
  double xs[NXs], us[NUs];
#pragma unroll
  for (int i = 0; i < NX; ++i) xs[i] = x[ph.x_state0 + i * (N + 1) + k];
#pragma unroll
  for (int j = 0; j < NU; ++j) us[j] = x[ph.x_control0 + j * N + k];
  double tk = tk0;
  // role -> perturbation pair (a, b); a = -1: base, b = -1: single
  int a = -1, b = -1, kind = 0, dst0 = 0, dst1 = 0;
  if (role >= 1 && role <= NV) a = role - 1;
  if (role > NV && role < NR) {
    const HessPairDev pr = Hp.pairs[hp.pair0 + (role - NV - 1)];
    a = pr.a; b = pr.b; kind = pr.kind; dst0 = pr.dst0; dst1 = pr.dst1;
  }
  // h = tol (1+|v|) of the UNPERTURBED value; a == b adds h twice: (v+h)+h, LpHessian.cpp:1268-1282
  double ha = 1.0, hb = 1.0;
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const double hi = K.tol * (1 + fabs(xs[i]));
    if (a == i) { ha = hi; xs[i] += hi; }
    if (b == i) { hb = hi; xs[i] += hi; }
  }
#pragma unroll
  for (int j = 0; j < NU; ++j) {
    const double hj = K.tol * (1 + fabs(us[j]));
    if (a == NX + j) { ha = hj; us[j] += hj; }
    if (b == NX + j) { hb = hj; us[j] += hj; }
  }
  {
    const double ht = K.tol * (1 + fabs(tk0));
    if (a == NX + NU) { ha = ht; tk += ht; }
    if (b == NX + NU) { hb = ht; tk += ht; }
  }
  double F[NF];
  {
    double cp[NCs];
    Prob::dae(ph.phase_num, tk, xs, us, c, F, cp);
#pragma unroll
    for (int j = 0; j < NC; ++j) F[NX + j] = cp[j];
    F[NX + NC] = Prob::lagrange(ph.phase_num, tk, xs, us, c);
  }
  if (act) {
#pragma unroll
    for (int o = 0; o < NF; ++o) lds[(role * NF + o) * TH + kk] = F[o];
  }
  __syncthreads();
  if (!act || role <= NV || kind == 0) return;
  // ---- combine: ((tf-t0)/2)(sigma w L_ab - sum lam f_ab) + sum mu c_ab   (LpHessian.cpp:119-129) ----
  const double den = ha * hb;
  const double* F0 = lds + kk;
  const double* Fa = lds + ((1 + a) * NF) * TH + kk;
  const double* Fb = lds + ((1 + b) * NF) * TH + kk;
  double sd = 0.0, sp = 0.0;
#pragma unroll
  for (int o = 0; o < NX; ++o) {
    const double hh = (F[o] - Fa[o * TH] - Fb[o * TH] + F0[o * TH]) / den;
    const double term = lam[o * N + k] * hh;
    sd = (o == 0) ? term : sd + term;
  }
#pragma unroll
  for (int o = 0; o < NC; ++o) {
    const double hh = (F[NX + o] - Fa[(NX + o) * TH] - Fb[(NX + o) * TH] + F0[(NX + o) * TH]) / den;
    const double term = lam[(NX + o) * N + k] * hh;
    sp = (o == 0) ? term : sp + term;
  }
  const double hL = (F[NX + NC] - Fa[(NX + NC) * TH] - Fb[(NX + NC) * TH] + F0[(NX + NC) * TH]) / den;
  const double XI = (tf - t0) / 2.0 * ((sigma * wq) * hL - sd) + sp;
  if (kind == 1) {
    hv[dst0 + k] = XI;
    return;
  }
  // ---- t0/tf rows: first-derivative pieces of variable b (:159-218).  Finite differences reuse the single
  //      perturbations already in LDS ((F_b - F_0)/h_b is exactly LpFDderive's formula) ----
  double D1;
  {
    double sdd = 0.0, dL;
    if constexpr (AN) {
      double xs0[NXs], us0[NUs], df[NXs], dc[NCs];
#pragma unroll
      for (int i = 0; i < NX; ++i) xs0[i] = x[ph.x_state0 + i * (N + 1) + k];
#pragma unroll
      for (int j = 0; j < NU; ++j) us0[j] = x[ph.x_control0 + j * N + k];
      Prob::dae_jac_col(ph.phase_num, b, tk0, xs0, us0, c, df, dc);
#pragma unroll
      for (int o = 0; o < NX; ++o) {
        const double term = lam[o * N + k] * df[o];
        sdd = (o == 0) ? term : sdd + term;
      }
      dL = Prob::lagrange_grad_col(ph.phase_num, b, tk0, xs0, us0, c);
    } else {
#pragma unroll
      for (int o = 0; o < NX; ++o) {
        const double term = lam[o * N + k] * ((Fb[o * TH] - F0[o * TH]) / hb);
        sdd = (o == 0) ? term : sdd + term;
      }
      dL = (Fb[(NX + NC) * TH] - F0[(NX + NC) * TH]) / hb;
    }
    D1 = sdd - (sigma * wq) * dL;
  }
  const double ta = (1 - tau) / 2.0, tb = (1 + tau) / 2.0;
  if (kind == 2) {
    hv[dst0 + k] = 0.5 * D1 + ta * XI;
    hv[dst1 + k] = -0.5 * D1 + tb * XI;
  } else {   // (t,t): per-node terms of the three dot products, reduced by rpm_hess_tt_kernel
    tmp[k] = ta * (D1 + ta * XI);
    tmp[N + k] = tb * (-D1 + tb * XI);
    tmp[2 * N + k] = 0.5 * ((tb - ta) * D1) + ta * (tb * XI);
  }
}

// t0t0, tftf, tft0 scalars: fixed-shape tree sums of the per-node terms (deterministic; the reference sums in
// Armadillo's dot order, so these three entries agree to rounding, not bit for bit)
__global__ void rpm_hess_tt_kernel(const KParams K, const HParams Hp, const double* __restrict__ tmp_all,
                                   double* __restrict__ hv_all) {
  __shared__ double red[3][256];
  const int p = blockIdx.x, inst = blockIdx.y, tid = threadIdx.x;
  const HessPhaseDev hp = Hp.phases[p];
  const int N = K.phases[p].N;
  const double* tmp = tmp_all + size_t(inst) * Hp.tmp_len + hp.tt_tmp;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (int k = tid; k < N; k += 256) {
    s0 += tmp[k];
    s1 += tmp[N + k];
    s2 += tmp[2 * N + k];
  }
  red[0][tid] = s0; red[1][tid] = s1; red[2][tid] = s2;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) {
      red[0][tid] += red[0][tid + s];
      red[1][tid] += red[1][tid + s];
      red[2][tid] += red[2][tid + s];
    }
    __syncthreads();
  }
  if (tid == 0) {
    double* hv = hv_all + size_t(inst) * Hp.nnz_h + hp.v0;
    hv[hp.tt_dst[0]] = red[0][0];   // t0t0
    hv[hp.tt_dst[2]] = red[1][0];   // tftf
    hv[hp.tt_dst[1]] = red[2][0];   // tft0
  }
}

// E-part (events + Mayer, LpHessian.cpp:1553-1983, assembled :290-330) and linkage entries (:1020-1190, :2163-2367):
// one thread per stored entry, four evaluations each (base, a, b, a+b).
template <class Prob>
__global__ void rpm_hess_end_kernel(const KParams K, const HParams Hp, const double* __restrict__ xall, const double sigma,
                                    const double* __restrict__ lam_all, double* __restrict__ hv_all) {
  constexpr int NX = Prob::NX;
  constexpr int NE = Prob::NE_MAX > 0 ? Prob::NE_MAX : 1, NL = Prob::NLINK_MAX > 0 ? Prob::NLINK_MAX : 1;
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  const int inst = blockIdx.y;
  const double* __restrict__ x = xall + size_t(inst) * K.n;
  const double* __restrict__ lam = lam_all + size_t(inst) * K.m;
  double* __restrict__ hv = hv_all + size_t(inst) * Hp.nnz_h;
  const double* c = K.consts;
  if (e < Hp.n_ends) {
    const HessEndDev en = Hp.ends[e];
    const PhaseDev ph = K.phases[en.phase];
    double x0[NX], xf[NX];
    for (int j = 0; j < NX; ++j) {
      x0[j] = x[ph.x_state0 + j * (ph.N + 1)];
      xf[j] = x[ph.x_state0 + j * (ph.N + 1) + ph.N];
    }
    const double t0 = x[ph.x_t0], tf = x[ph.x_t0 + 1];
    auto pert = [&](int v) -> double {
      double base = v < NX ? x0[v < NX ? v : 0] : (v < 2 * NX ? xf[v - NX] : (v == 2 * NX ? t0 : tf));
      return K.tol * (1 + fabs(base));
    };
    const double pa = pert(en.a), pb = pert(en.b), den = pert(en.da) * pert(en.db);
    double ev[4][NE], my[4];
    for (int q = 0; q < 4; ++q) {   // 0: base, 1: a, 2: b, 3: a then b
      double y0[NX], yf[NX], s0 = t0, sf = tf;
      for (int j = 0; j < NX; ++j) { y0[j] = x0[j]; yf[j] = xf[j]; }
      for (int w = 0; w < 2; ++w) {
        const bool on = (w == 0) ? (q == 1 || q == 3) : (q == 2 || q == 3);
        if (!on) continue;
        const int v = w == 0 ? en.a : en.b;
        const double hh = w == 0 ? pa : pb;
        for (int j = 0; j < NX; ++j) {
          if (v == j) y0[j] += hh;
          if (v == NX + j) yf[j] += hh;
        }
        if (v == 2 * NX) s0 += hh;
        if (v == 2 * NX + 1) sf += hh;
      }
      for (int i = 0; i < NE; ++i) ev[q][i] = 0.0;
      if (ph.ne > 0) Prob::event(ph.phase_num, s0, y0, sf, yf, c, ev[q]);
      my[q] = Prob::mayer(ph.phase_num, s0, y0, sf, yf, c);
    }
    const double hM = (my[3] - my[1] - my[2] + my[0]) / den;
    double v1 = 0.0, v2 = 0.0;   // accu(hEvents % event_lambda): two interleaved accumulators
    const double* lam_e = lam + ph.g0 + (NX + Prob::NC) * ph.N;
    int i = 0;
    for (; i + 1 < ph.ne; i += 2) {
      v1 += ((ev[3][i] - ev[1][i] - ev[2][i] + ev[0][i]) / (den * 1.0)) * lam_e[i];
      v2 += ((ev[3][i + 1] - ev[1][i + 1] - ev[2][i + 1] + ev[0][i + 1]) / (den * 1.0)) * lam_e[i + 1];
    }
    if (i < ph.ne) v1 += ((ev[3][i] - ev[1][i] - ev[2][i] + ev[0][i]) / (den * 1.0)) * lam_e[i];
    hv[Hp.phases[en.phase].v0 + en.dst] = sigma * hM + (v1 + v2);
  } else if (e < Hp.n_ends + Hp.n_links) {
    const HessLinkDev le = Hp.links[e - Hp.n_ends];
    const LinkDev lk = K.links[le.pair];
    const PhaseDev pl = K.phases[lk.left];
    const PhaseDev pr = K.phases[lk.right];
    double w0[2 * NX];
    for (int j = 0; j < NX; ++j) {
      w0[j] = x[pl.x_state0 + j * (pl.N + 1) + pl.N];
      w0[NX + j] = x[pr.x_state0 + j * (pr.N + 1)];
    }
    const double pa = K.tol * (1 + fabs(w0[le.a])), pb = K.tol * (1 + fabs(w0[le.b]));
    double lo[4][NL];
    for (int q = 0; q < 4; ++q) {
      double w[2 * NX];
      for (int j = 0; j < 2 * NX; ++j) w[j] = w0[j];
      if (q == 1 || q == 3) w[le.a] += pa;
      if (q == 2 || q == 3) w[le.b] += pb;
      for (int i = 0; i < NL; ++i) lo[q][i] = 0.0;
      Prob::link(lk.left + 1, lk.right + 1, w, w + NX, c, lk.nlink, lo[q]);
    }
    // link multipliers: the reference reads the FIRST pair's rows for every pair (link_indices are built
    // without advancing the offset, Core/LpBoundsChecker.cpp:240-244) — kept
    const double* lam_l = lam + K.links[0].g0;
    const double den = pa * pb;
    double v1 = 0.0, v2 = 0.0;
    int i = 0;
    for (; i + 1 < lk.nlink; i += 2) {
      v1 += ((lo[3][i] - lo[1][i] - lo[2][i] + lo[0][i]) / den) * lam_l[i];
      v2 += ((lo[3][i + 1] - lo[1][i + 1] - lo[2][i + 1] + lo[0][i + 1]) / den) * lam_l[i + 1];
    }
    if (i < lk.nlink) v1 += ((lo[3][i] - lo[1][i] - lo[2][i] + lo[0][i]) / den) * lam_l[i];
    hv[le.dst] = v1 + v2;
  }
}

// ------------------------------------------------------------------------------------------
// Mesh-error estimate: SolutionErrorChecker::CheckSolutionDiffError, Core/LpSolutionError.cpp:112-169.
// One workgroup per mesh interval.  Phase A interpolates the interval's states / controls onto its (n+1)-point LGR
// mesh (SolutionInterpolation, :46-108, rows of the tables built in rpm_mesh.cpp), phase B evaluates the dynamics
// there, phase C integrates them with the interval's integration matrix: X(start) + A f (:147).
template <class Prob>
__global__ void rpm_mesh_err_kernel(const KParams K, int phase, const double* __restrict__ x,
                                    const MeshIvDev* __restrict__ ivs, int n_iv, const double* __restrict__ Hs,
                                    const double* __restrict__ Ss, const int* __restrict__ hit_s,
                                    const double* __restrict__ Hc, const double* __restrict__ Sc,
                                    const int* __restrict__ hit_c, const double* __restrict__ A,
                                    const double* __restrict__ ttem, int rows, double* __restrict__ fine_state,
                                    double* __restrict__ integ) {
  constexpr int NX = Prob::NX, NU = Prob::NU, NC = Prob::NC;
  constexpr int NXs = NX > 0 ? NX : 1, NUs = NU > 0 ? NU : 1, NCs = NC > 0 ? NC : 1;
  extern __shared__ double mesh_sm[];
  const MeshIvDev v = ivs[blockIdx.x];
  const int n = v.n, n1 = n + 1;
  double* Xs = mesh_sm;            // [q * NX + s]
  double* Us = Xs + n1 * NX;       // [q * NU + j]
  double* Fs = Us + n1 * NU;       // [q * NX + s]
  const PhaseDev ph = K.phases[phase];
  const int N = ph.N, M = N + 1;
  const double t0 = x[ph.x_t0];
  const double tf = (x[ph.x_t0 + 1] - t0) * (1.0 + 1) / 2 + t0;   // result->time's last entry, Nlp2OPConverter.cpp:58
  for (int idx = threadIdx.x; idx < n1 * NX; idx += blockDim.x) {
    const int q = idx % n1, s = idx / n1;
    const double* col = x + ph.x_state0 + s * M + v.istart;
    const int hit = hit_s[v.q0 + q];
    double val;
    if (hit >= 0) {
      val = col[hit];
    } else {
      double acc = 0.0;
      for (int j = 0; j < n1; ++j) acc += Hs[v.hs + q + j * n1] * col[j];
      val = acc / Ss[v.q0 + q];
    }
    Xs[q * NX + s] = val;
    fine_state[(v.r0 + q) + size_t(s) * rows] = val;
  }
  for (int idx = threadIdx.x; idx < n1 * NU; idx += blockDim.x) {
    const int q = idx % n1, j = idx / n1;
    const double* col = x + ph.x_control0 + j * N + v.istart;
    const int hit = hit_c[v.q0 + q];
    double val;
    if (hit >= 0) {
      val = col[hit];
    } else {
      double acc = 0.0;
      for (int c = 0; c < n; ++c) acc += Hc[v.hc + q + c * n1] * col[c];
      val = acc / Sc[v.q0 + q];
    }
    Us[q * NU + j] = val;
  }
  __syncthreads();
  const double half = (tf - t0) / 2;
  for (int q = threadIdx.x; q < n1; q += blockDim.x) {
    double xs[NXs], us[NUs], f[NXs], cp[NCs];
#pragma unroll
    for (int s = 0; s < NX; ++s) xs[s] = Xs[q * NX + s];
#pragma unroll
    for (int j = 0; j < NU; ++j) us[j] = Us[q * NU + j];
    const double t = half * ttem[v.q0 + q] + half;   // t0 is not added, LpSolutionError.cpp:124
    Prob::dae(ph.phase_num, t, xs, us, K.consts, f, cp);
#pragma unroll
    for (int s = 0; s < NX; ++s) Fs[q * NX + s] = f[s] * ((tf - t0) / 2.0);
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < n1 * NX; idx += blockDim.x) {
    const int r = idx % n1, s = idx / n1;
    double acc = 0.0;
    for (int c = 0; c < n1; ++c) acc += A[v.a + r + c * n1] * Fs[c * NX + s];
    integ[(1 + v.r0 + r) + size_t(s) * rows] = (0.0 + 1.0 * Xs[s]) + acc;
  }
  if (blockIdx.x == 0)
    for (int s = threadIdx.x; s < NX; s += blockDim.x) integ[size_t(s) * rows] = Xs[s];
  if (blockIdx.x == n_iv - 1)
    for (int s = threadIdx.x; s < NX; s += blockDim.x)
      fine_state[(rows - 1) + size_t(s) * rows] = x[ph.x_state0 + s * M + N];
}

// relative_error(:, s) = |integrated - interpolated| / (1 + max(interpolated(:, s))), one workgroup per state (:148-157)
__global__ void rpm_mesh_rel_kernel(int rows, const double* __restrict__ fine_state, const double* __restrict__ integ,
                                    double* __restrict__ rel) {
  __shared__ double red[256];
  const double* col = fine_state + size_t(blockIdx.x) * rows;
  double mx = col[0];
  for (int r = threadIdx.x; r < rows; r += blockDim.x) mx = fmax(mx, col[r]);
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (threadIdx.x < w) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + w]);
    __syncthreads();
  }
  const double den = 1 + red[0];
  for (int r = threadIdx.x; r < rows; r += blockDim.x)
    rel[r + size_t(blockIdx.x) * rows] = fabs(integ[r + size_t(blockIdx.x) * rows] - col[r]) / den;
}

// ------------------------------------------------------------------------------------------
// Solution extraction: Nlp2OpConverter::Nlp2OpControl, Core/Nlp2OPConverter.cpp:13-196.
// Runs once per mesh after the NLP solve, not per iteration.
// rpm_post_spline_kernel: value at tau = +1 of the natural cubic spline through (tau_k, y_k), one thread per column
// (LpGuessChecker::spline_interpolation, Core/LpGuessChecker.cpp:208-270, specialised to the last interval: only the
// forward recurrence's final z is needed because c[n-1] = 0).
__global__ void rpm_post_spline_kernel(int N, const double* __restrict__ tau, const double* __restrict__ cols, int ncols,
                                       double scale_num, double scale_den, const double* __restrict__ w,
                                       double* __restrict__ out) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= ncols) return;
  const double* y = cols + size_t(col) * N;
  // optional scaling y_k -> scale_num * (1/w_k) * y_k / scale_den  (path multipliers, Nlp2OPConverter.cpp:92)
  auto Y = [&](int k) -> double { return w ? scale_num * ((1 / w[k]) * y[k]) / scale_den : y[k]; };
  double mu = 0.0, z = 0.0;
  for (int i = 1; i < N - 1; ++i) {
    const double him1 = tau[i] - tau[i - 1], hi = tau[i + 1] - tau[i];
    const double alpha = 3.0 / hi * (Y(i + 1) - Y(i)) - 3.0 / him1 * (Y(i) - Y(i - 1));
    const double li = 2 * (tau[i + 1] - tau[i - 1]) - him1 * mu;
    mu = hi / li;
    z = (alpha - him1 * z) / li;
  }
  const double d2l = (N - 2 >= 1) ? 2 * z : 0.0;   // c[n-2] = z[n-2] - mu[n-2]*c[n-1], doubled for interior knots
  const double h = tau[N - 1] - tau[N - 2];
  const double A = (tau[N - 1] - 1.0) / h, B = (1.0 - tau[N - 2]) / h;
  const double Cc = (pow(A, 3.0) - A) * (h * h) / 6.0, Dd = (pow(B, 3.0) - B) * (h * h) / 6.0;
  out[col] = A * Y(N - 2) + B * Y(N - 1) + Cc * d2l + Dd * 0.0;
}

template <class Prob>
__global__ void rpm_post_kernel(const KParams K, int phase, const double* __restrict__ x, const double* __restrict__ lam,
                                const double* __restrict__ u_end, const double* __restrict__ pm_end,
                                double* __restrict__ o_time, double* __restrict__ o_state, double* __restrict__ o_control,
                                double* __restrict__ o_costate, double* __restrict__ o_pathmult,
                                double* __restrict__ o_ham, double* __restrict__ o_lag, double* __restrict__ o_mayer) {
  constexpr int NX = Prob::NX, NU = Prob::NU, NC = Prob::NC;
  constexpr int NXs = NX > 0 ? NX : 1, NUs = NU > 0 ? NU : 1, NCs = NC > 0 ? NC : 1;
  const PhaseDev ph = K.phases[phase];
  const int N = ph.N, M = N + 1;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= M) return;
  const double t0 = x[ph.x_t0], tf = x[ph.x_t0 + 1];
  const double tau = k < N ? K.points[ph.node0 + k] : 1.0;
  const double t = (tf - t0) * (tau + 1) / 2 + t0;                       // :49
  o_time[k] = t;
  double xs[NXs], us[NUs], cst[NXs];
#pragma unroll
  for (int s = 0; s < NX; ++s) {
    xs[s] = x[ph.x_state0 + s * M + k];
    o_state[s * M + k] = xs[s];
  }
#pragma unroll
  for (int j = 0; j < NU; ++j) {
    us[j] = k < N ? x[ph.x_control0 + j * N + k] : u_end[j];             // :53-64
    o_control[j * M + k] = us[j];
  }
  const double* lp = lam + ph.g0;                                       // this phase's multipliers, :73
#pragma unroll
  for (int s = 0; s < NX; ++s) {
    if (k < N) {
      cst[s] = -((1 / K.weights[ph.node0 + k]) * lp[s * N + k]);         // -(W^-1 lambda), :75-79
    } else {
      // -trans(D(:,N)) * lambda: only the rows of the last mesh interval reach the last column
      const NodeDev last = K.nodes[ph.node0 + N - 1];
      double acc = 0.0;
      for (int r = last.dcol0; r < N; ++r) {
        const NodeDev nr = K.nodes[ph.node0 + r];
        acc += K.dvals[nr.drow_off + nr.dlen - 1] * lp[s * N + r];
      }
      cst[s] = -acc;
    }
    o_costate[s * M + k] = cst[s];
  }
#pragma unroll
  for (int j = 0; j < NC; ++j)   // lambda WITHOUT the phase offset, exactly as Nlp2OPConverter.cpp:88 reads it
    o_pathmult[j * M + k] = k < N ? 2 * ((1 / K.weights[ph.node0 + k]) * lam[N * NX + j * N + k]) / (tf - t0) : pm_end[j];
  double f[NXs], cp[NCs];
  Prob::dae(ph.phase_num, t, xs, us, K.consts, f, cp);
  const double L = Prob::lagrange(ph.phase_num, t, xs, us, K.consts);
  double sum = 0.0;
#pragma unroll
  for (int s = 0; s < NX; ++s) {
    const double term = cst[s] * f[s];
    sum = (s == 0) ? term : sum + term;
  }
  o_ham[k] = L + sum;                                                    // :146
  o_lag[k] = L;
  if (k == 0) {
    double x0[NXs], xf[NXs];
#pragma unroll
    for (int s = 0; s < NX; ++s) {
      x0[s] = x[ph.x_state0 + s * M];
      xf[s] = x[ph.x_state0 + s * M + N];
    }
    o_mayer[0] = Prob::mayer(ph.phase_num, t0, x0, tf, xf, K.consts);
  }
}

// lagrange_cost = (tf-t0) * (w . L[0..N-1]) / 2  (:134), fixed-tree sum
__global__ void rpm_post_cost_kernel(const KParams K, int phase, const double* __restrict__ x,
                                     const double* __restrict__ lag, double* __restrict__ out) {
  __shared__ double red[256];
  const PhaseDev ph = K.phases[phase];
  const int tid = threadIdx.x;
  double s = 0.0;
  for (int k = tid; k < ph.N; k += 256) s += K.weights[ph.node0 + k] * lag[k];
  red[tid] = s;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) red[tid] += red[tid + st];
    __syncthreads();
  }
  if (tid == 0) out[0] = (x[ph.x_t0 + 1] - x[ph.x_t0]) * red[0] / 2.0;
}

// ------------------------------------------------------------------------------------------
// problem registry
template <class F>
static bool with_problem(int id, F&& fn) {
  switch (id) {
    case RPM_PROBLEM_LAUNCH: fn(LaunchProblem{}); return true;
    case RPM_PROBLEM_HYPERSENSITIVE: fn(HypersensitiveProblem{}); return true;
    case RPM_PROBLEM_BRYSON_DENHAM: fn(BrysonDenhamProblem{}); return true;
    case RPM_PROBLEM_BRACHISTOCHRONE: fn(BrachistochroneProblem{}); return true;
    case RPM_PROBLEM_MIN_TIME_CLIMB: fn(MinTimeClimbProblem{}); return true;
    case RPM_PROBLEM_QUADROTOR: fn(QuadrotorProblem{}); return true;
  }
  return false;
}

bool problem_dims(int id, ProblemDims* out) {
  return with_problem(id, [&](auto prob) {
    using P = decltype(prob);
    *out = ProblemDims{P::NX, P::NU, P::NC, P::NE_MAX, P::NLINK_MAX, P::NCONST, P::HAS_ANALYTIC};
  });
}

template <class T>
static hipError_t upload(T** dst, const std::vector<T>& src) {
  const size_t bytes = (src.size() ? src.size() : 1) * sizeof(T);
  hipError_t s = hipMalloc(reinterpret_cast<void**>(dst), bytes);
  if (s != hipSuccess) return s;
  if (!src.empty()) s = hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice);
  return s;
}

void device_destroy(Engine& e) {
  Device* d = e.dev;
  if (!d) return;
  (void)hipSetDevice(d->device_id);
#ifdef RPM_DIAG
  if (d->kp.trace) {   // dump the last launch's per-workgroup timestamps
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(d->trace_words);
    (void)hipMemcpy(h.data(), d->kp.trace, d->trace_words * 8, hipMemcpyDeviceToHost);
    if (FILE* f = std::fopen(getenv("RPM_DIAG_TRACE"), "wb")) {
      std::fwrite(h.data(), 8, h.size(), f);
      std::fclose(f);
    }
    (void)hipFree(d->kp.trace);
  }
#endif
  void* ptrs[] = {d->d_phases, d->d_tiles, d->d_tasks, d->d_nodes, d->d_points, d->d_weights, d->d_diag,
                  d->d_dvals, d->d_doff, d->d_consts, d->d_alin_v, d->d_links, d->d_alin_j, d->d_x, d->d_g,
                  d->d_values, d->d_grad, d->d_obj, d->d_lambda, d->d_hess, d->d_partial, d->d_hpairs, d->d_hphases,
                  d->d_hends, d->d_hlinks, d->d_htiles, d->d_htmp};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (d->d_flag) (void)hipFree(d->d_flag);
  if (d->d_flags2) (void)hipFree(d->d_flags2);
  if (d->h_flags2) (void)hipHostFree(d->h_flags2);
  for (auto& p : d->pinned) (void)hipHostUnregister(const_cast<void*>(p.first));
  (void)hipGetLastError();   // a buffer the caller already freed makes the unregister fail: not an error of ours
  for (auto& row : d->segtab)
    for (auto& t : row)
      if (t.ptr) (void)hipFree(t.ptr);
  if (d->stream) (void)hipStreamDestroy(d->stream);
  delete d;
  e.dev = nullptr;
}

static size_t tile_lds_doubles(const Engine& e, int NX, int NU, int NC) {
  size_t n = size_t(NX) * e.max_span + size_t(NU) * e.tile_nodes + e.max_drow + size_t(NX + NC) * e.tile_nodes + size_t(NX) * e.tile_nodes;
  return n < 64 ? 64 : n;
}

int device_init(Engine& e, int device_id) {
  if (e.dev) {
    if (e.dev->device_id == device_id) return RPM_OK;
    device_destroy(e);
  }
  int count = 0;
  hipError_t s = hipGetDeviceCount(&count);
  if (s != hipSuccess || count <= 0) {
    e.err = "no HIP device available (this engine has no CPU fallback)";
    return RPM_E_DEVICE;
  }
  if (device_id < 0 || device_id >= count) {
    e.err = "device id out of range";
    return RPM_E_DEVICE;
  }
  HIP_TRY(e, hipSetDevice(device_id));
  Device* d = new Device();
  e.dev = d;
  d->device_id = device_id;
  HIP_TRY(e, hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
  HIP_TRY(e, upload(&d->d_phases, e.phd));
  {
    std::vector<TileDev> mine;
    for (int id : e.my_tiles) mine.push_back(e.tiles[id]);
    HIP_TRY(e, upload(&d->d_tiles, mine));
  }
  HIP_TRY(e, upload(&d->d_tasks, e.tasks));
  HIP_TRY(e, upload(&d->d_nodes, e.nodes));
  HIP_TRY(e, upload(&d->d_points, e.points));
  HIP_TRY(e, upload(&d->d_weights, e.weights));
  HIP_TRY(e, upload(&d->d_diag, e.diag));
  HIP_TRY(e, upload(&d->d_dvals, e.dvals));
  HIP_TRY(e, upload(&d->d_doff, e.doff_vals));
  HIP_TRY(e, upload(&d->d_consts, e.consts));
  HIP_TRY(e, upload(&d->d_links, e.links));
  HIP_TRY(e, upload(&d->d_alin_j, e.alin_j));
  HIP_TRY(e, upload(&d->d_alin_v, e.alin_v));
  const size_t B = size_t(e.n_instances);
  HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&d->d_x), B * e.n * sizeof(double)));
  HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&d->d_g), B * e.m * sizeof(double)));
  HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&d->d_values), B * size_t(e.nnz_jac) * sizeof(double)));
  HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&d->d_grad), B * e.n * sizeof(double)));
  HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&d->d_obj), B * sizeof(double)));
  HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&d->d_partial), B * e.P * sizeof(double)));
  HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&d->d_lambda), B * e.m * sizeof(double)));
  HIP_TRY(e, hipMemset(d->d_grad, 0, B * e.n * sizeof(double)));
  KParams& k = d->kp;
  k.phases = d->d_phases;
  k.tiles = d->d_tiles;
  k.n_my_tiles = int(e.my_tiles.size());
  k.nodes = d->d_nodes;
  k.points = d->d_points;
  k.weights = d->d_weights;
  k.diag = d->d_diag;
  k.dvals = d->d_dvals;
  k.doff_vals = d->d_doff;
  k.consts = d->d_consts;
  k.links = d->d_links;
  k.alin_j = d->d_alin_j;
  k.alin_v = d->d_alin_v;
  k.tol = e.fd_tol;
  k.P = e.P;
  k.L = e.L;
  k.n = e.n;
  k.m = e.m;
  k.m_nl = e.m_nl;
  k.nnz = e.nnz_jac;
  k.nnz_nl = e.nnz_nl;
  k.nnz_lin = e.nnz_lin;
  k.nnz_const = e.nnz_const;
  k.max_span = e.max_span;
  k.max_drow = e.max_drow;
  k.max_cshare = 0;
  for (const TileDev& t : e.tiles) k.max_cshare = t.c_cnt > k.max_cshare ? t.c_cnt : k.max_cshare;
  const bool sharded = e.shard_mode == RPM_SHARD_INTERVALS && e.shard_world > 1;
  k.tasks = d->d_tasks;
  k.n_tasks = (!sharded || e.shard_rank == 0) ? int(e.tasks.size()) : 0;  // rank 0 owns the endpoint rows
  k.diag_mask = 0;
  k.trace = nullptr;
#ifdef RPM_DIAG
  if (getenv("RPM_DIAG_TRACE")) {
    size_t words = size_t(e.tiles.size() * 2 + e.tasks.size() + 64) * size_t(e.n_instances) * 8;
    if (words < size_t(4096) * 64) words = size_t(4096) * 64;   // the pipelined kernel traces 64 words per half-workgroup
    if (hipMalloc(reinterpret_cast<void**>(&k.trace), words * 8) == hipSuccess) (void)hipMemset(k.trace, 0, words * 8);
    d->trace_words = words;
  }
  if (const char* dm = getenv("RPM_DIAG_MASK")) k.diag_mask = atoi(dm);
#endif
  ProblemDims pd;
  problem_dims(e.problem_id, &pd);
  d->lds_bytes = tile_lds_doubles(e, pd.nx, pd.nu, pd.nc) * sizeof(double);
  if (d->lds_bytes > 160 * 1024) {
    e.err = "mesh interval too large for the LDS-staged D tile (reduce nodes per interval)";
    return RPM_E_UNSUPPORTED;
  }
  {   // rpm_tile_pl_kernel: persistent workgroups of RG compute + 2 DMA waves; as many per CU as the occupancy
      // calculator grants the full (g + Jacobian) variant
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device_id) != hipSuccess || ncu <= 0) ncu = 256;
    const int max_c = d->kp.max_cshare;
    const size_t stage = size_t(PL_REC + 2) + size_t(pd.nx) * e.max_span + size_t(pd.nu) * 64 + e.max_drow + 4 * 64 + max_c;
    int per_cu = 0;
    with_problem(e.problem_id, [&](auto prob) {
      using P = decltype(prob);
      constexpr PlShape S = pl_shape(P::NX + P::NU + 2);
      d->pl_lds = S.NH * (2 * stage + size_t(pd.nx + pd.nc) * 64 + 2) * sizeof(double);
      if (d->pl_lds > 160 * 1024) return;
      auto kern = rpm_tile_pl_kernel<P, S.NH, S.RG, S.NDMA, true, true, false>;
      if (d->pl_lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  int(d->pl_lds));
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, S.NH * 64 * (S.RG + S.NDMA), d->pl_lds) != hipSuccess)
        per_cu = 0;
      (void)hipGetLastError();
      per_cu *= S.NH;   // resident halves per CU
    });
    d->pl_slots = per_cu * ncu;
    d->pl_ok = e.role_looped && e.tile_nodes == 64 && max_c <= PL_CMAX && max_c >= 2 && 2 * pd.nx + 3 <= 64 &&
               per_cu >= 1;
  }
  return RPM_OK;
}

// ------------------------------------------------------------------------------------------
template <class Prob, int T, bool WG, bool WJ, bool AN, bool DXM = false>
static hipError_t launch_tile_inst(const Engine& e, const double* dx, double* dg, double* dv, hipStream_t st) {
  constexpr int R = WJ ? Prob::NX + Prob::NU + 2 : (Prob::NX > 0 ? Prob::NX : 1);
  int threads = T * R;
  threads = (threads + 63) / 64 * 64;
  if (threads < 64) threads = 64;
  const Device& d = *e.dev;
  auto kern = rpm_tile_kernel<Prob, T, WG, WJ, AN, DXM>;
  if (d.lds_bytes > 64 * 1024) {
    hipError_t s = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, int(d.lds_bytes));
    if (s != hipSuccess) return s;
  }
  dim3 grid(unsigned(d.kp.n_my_tiles + d.kp.n_tasks), unsigned(e.n_instances));
  hipLaunchKernelGGL(kern, grid, dim3(threads), d.lds_bytes, st, d.kp, dx, dg, dv);
  return hipGetLastError();
}

template <class Prob, int T, int RG, bool WG, bool WJ, bool AN>
static hipError_t launch_tile_rl(const Engine& e, const double* dx, double* dg, double* dv, hipStream_t st) {
  const Device& d = *e.dev;
  auto kern = rpm_tile_rl_kernel<Prob, T, RG, WG, WJ, AN>;
  if (d.lds_bytes > 64 * 1024) {
    hipError_t s = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       int(d.lds_bytes));
    if (s != hipSuccess) return s;
  }
  dim3 grid(unsigned(d.kp.n_my_tiles + d.kp.n_tasks), unsigned(e.n_instances));
  hipLaunchKernelGGL(kern, grid, dim3(T * RG), d.lds_bytes, st, d.kp, dx, dg, dv);
  return hipGetLastError();
}

template <class Prob, bool WG, bool WJ, bool AN>
static hipError_t launch_tile_pl(const Engine& e, const double* dx, double* dg, double* dv, hipStream_t st) {
  const Device& d = *e.dev;
  constexpr PlShape S = pl_shape(Prob::NX + Prob::NU + 2);
  auto kern = rpm_tile_pl_kernel<Prob, S.NH, S.RG, S.NDMA, WG, WJ, AN>;
  if (d.pl_lds > 64 * 1024) {
    hipError_t s = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       int(d.pl_lds));
    if (s != hipSuccess) return s;
  }
  const long long W = (long long)d.kp.n_my_tiles * e.n_instances;
  const long long halves = W < d.pl_slots ? W : d.pl_slots;   // pl_slots: resident halves (occupancy query)
  hipLaunchKernelGGL(kern, dim3(unsigned((halves + S.NH - 1) / S.NH)), dim3(S.NH * 64 * (S.RG + S.NDMA)), d.pl_lds, st, d.kp,
                     e.n_instances, dx, dg, dv);
  return hipGetLastError();
}

// the pipelined kernel pays off once every resident workgroup has at least two tiles to walk
static bool use_pipeline(const Engine& e) {
  const Device& d = *e.dev;
  if (!d.pl_ok || e.opt_pipeline == 0 || d.kp.n_my_tiles <= 0) return false;
  if (e.opt_pipeline == 1) return true;
  // measured on the metric problem: it wins when every half-workgroup walks >= 2 tiles, or when the tiles fill the
  // resident halves exactly once (the prefetch then hides nothing, but the DMA waves still take the constant block);
  // in between, the second pass over a partly filled chip loses to the role-looped kernel
  const long long W = (long long)d.kp.n_my_tiles * e.n_instances;
  return W >= 2LL * d.pl_slots || (W <= d.pl_slots && 4 * W >= 3LL * d.pl_slots);
}

int dev_pipeline_active(const Engine& e) { return e.dev && e.role_looped && e.opt_dx_mode == 0 && use_pipeline(e) ? 1 : 0; }

template <class Prob, int T>
static hipError_t launch_tile_T(const Engine& e, bool wg, bool wj, const double* dx, double* dg, double* dv,
                                hipStream_t st) {
  if (e.role_looped && T == 64 && e.opt_dx_mode == 0 && use_pipeline(e)) {
    const bool an_pl = e.first_derive == RPM_DERIVE_ANALYTIC;
    if constexpr (Prob::HAS_ANALYTIC) {
      if (an_pl) {
        if (wg && wj) return launch_tile_pl<Prob, true, true, true>(e, dx, dg, dv, st);
        if (wj) return launch_tile_pl<Prob, false, true, true>(e, dx, dg, dv, st);
      }
    }
    if (wg && wj) return launch_tile_pl<Prob, true, true, false>(e, dx, dg, dv, st);
    if (wj) return launch_tile_pl<Prob, false, true, false>(e, dx, dg, dv, st);
    return launch_tile_pl<Prob, true, false, false>(e, dx, dg, dv, st);
  }
  if (e.role_looped && T == 64 && e.opt_dx_mode == 0) {   // throughput layout (see rpm_tile_rl_kernel)
    const bool an_rl = e.first_derive == RPM_DERIVE_ANALYTIC;
    if constexpr (Prob::HAS_ANALYTIC) {
      if (an_rl) {
        if (wg && wj) return launch_tile_rl<Prob, 64, 4, true, true, true>(e, dx, dg, dv, st);
        if (wj) return launch_tile_rl<Prob, 64, 4, false, true, true>(e, dx, dg, dv, st);
      }
    }
    if (wg && wj) return launch_tile_rl<Prob, 64, 4, true, true, false>(e, dx, dg, dv, st);
    if (wj) return launch_tile_rl<Prob, 64, 4, false, true, false>(e, dx, dg, dv, st);
    return launch_tile_rl<Prob, 64, 4, true, false, false>(e, dx, dg, dv, st);
  }
  const bool an = e.first_derive == RPM_DERIVE_ANALYTIC;
  if constexpr (Prob::HAS_ANALYTIC) {
    if (an) {
      if (wg && wj) return launch_tile_inst<Prob, T, true, true, true>(e, dx, dg, dv, st);
      if (wj) return launch_tile_inst<Prob, T, false, true, true>(e, dx, dg, dv, st);
    }
  }
  if (e.opt_dx_mode == 1) {   // MFMA D.X (finite-difference derivative mode)
    if (wg && wj) return launch_tile_inst<Prob, T, true, true, false, true>(e, dx, dg, dv, st);
    if (wg) return launch_tile_inst<Prob, T, true, false, false, true>(e, dx, dg, dv, st);
  }
  if (wg && wj) return launch_tile_inst<Prob, T, true, true, false>(e, dx, dg, dv, st);
  if (wj) return launch_tile_inst<Prob, T, false, true, false>(e, dx, dg, dv, st);
  return launch_tile_inst<Prob, T, true, false, false>(e, dx, dg, dv, st);
}

// flags: bit0 = g, bit1 = jacobian values
int dev_eval_cons(Engine& e, const double* d_x, double* d_g, double* d_values, int flags, void* stream) {
  if (!e.dev) {
    int rc = device_init(e, 0);
    if (rc) return rc;
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool wg = flags & 1, wj = flags & 2;
  hipError_t s = hipErrorInvalidValue;
  with_problem(e.problem_id, [&](auto prob) {
    using P = decltype(prob);
    switch (e.tile_nodes) {
      case 64: s = launch_tile_T<P, 64>(e, wg, wj, d_x, d_g, d_values, st); break;
      case 32: s = launch_tile_T<P, 32>(e, wg, wj, d_x, d_g, d_values, st); break;
      default: s = launch_tile_T<P, 16>(e, wg, wj, d_x, d_g, d_values, st); break;
    }
  });
  if (s != hipSuccess) {
    e.err = std::string("rpm_tile_kernel launch: ") + hipGetErrorString(s);
    return RPM_E_DEVICE;
  }
  return RPM_OK;
}

int dev_eval_obj(Engine& e, const double* d_x, double* d_obj, double* d_grad, void* stream) {
  if (!e.dev) {
    int rc = device_init(e, 0);
    if (rc) return rc;
  }
  Device& d = *e.dev;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool an = e.first_derive == RPM_DERIVE_ANALYTIC;
  hipError_t s = hipSuccess;
  with_problem(e.problem_id, [&](auto prob) {
    using P = decltype(prob);
    dim3 grid(unsigned(e.P), unsigned(e.n_instances));
    if (d_grad) {
      bool done = false;
      if constexpr (P::HAS_ANALYTIC) {
        if (an) {
          hipLaunchKernelGGL((rpm_obj_kernel<P, true, true>), grid, dim3(256), 0, st, d.kp, d_x, d_obj, d_grad, d.d_partial);
          done = true;
        }
      }
      if (!done)
        hipLaunchKernelGGL((rpm_obj_kernel<P, true, false>), grid, dim3(256), 0, st, d.kp, d_x, d_obj, d_grad, d.d_partial);
    } else {
      hipLaunchKernelGGL((rpm_obj_kernel<P, false, false>), grid, dim3(256), 0, st, d.kp, d_x, d_obj, d_grad, d.d_partial);
    }
    s = hipGetLastError();
  });
  if (s == hipSuccess && d_obj) {
    const int B = e.n_instances;
    const int thr = B < 256 ? B : 256;
    hipLaunchKernelGGL(rpm_obj_sum_kernel, dim3(unsigned((B + thr - 1) / thr)), dim3(unsigned(thr)), 0, st, e.P, B,
                       d.d_partial, d_obj);
    s = hipGetLastError();
  }
  if (s != hipSuccess) {
    e.err = std::string("rpm_obj_kernel launch: ") + hipGetErrorString(s);
    return RPM_E_DEVICE;
  }
  return RPM_OK;
}

// ---- non-finite detection on the device (the host-pointer path reports NaN/Inf to Ipopt) ---------
__global__ void rpm_finite_kernel(const double* __restrict__ v, size_t n, int* __restrict__ flag) {
  bool bad = false;
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += size_t(gridDim.x) * blockDim.x)
    bad |= !isfinite(v[i]);
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

// 1 if dev[0..count) contains a NaN/Inf (checked on the device, 4 bytes cross PCIe), 0 if not, <0 on error
int dev_nonfinite(Engine& e, const double* dev, size_t count) {
  Device& d = *e.dev;
  if (!d.d_flag) {
    if (hipMalloc(reinterpret_cast<void**>(&d.d_flag), sizeof(int)) != hipSuccess) return -1;
  }
  if (hipMemsetAsync(d.d_flag, 0, sizeof(int), d.stream) != hipSuccess) return -1;
  unsigned blocks = unsigned((count + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(rpm_finite_kernel, dim3(blocks), dim3(256), 0, d.stream, dev, count, d.d_flag);
  int h = 0;
  if (hipMemcpyAsync(&h, d.d_flag, sizeof(int), hipMemcpyDeviceToHost, d.stream) != hipSuccess) return -1;
  if (hipStreamSynchronize(d.stream) != hipSuccess) return -1;
  return h;
}

// The same check without its own round trip: zero flag word `slot` (0 or 1) and scan dev[0..count) on the engine's
// stream; dev_flags_fetch queues the copy of both words into page-locked host memory.  The caller synchronises once,
// together with its result download, and reads dev_flag_value.
int dev_nonfinite_enqueue(Engine& e, const double* dev, size_t count, int slot) {
  Device& d = *e.dev;
  if (!d.d_flags2) {
    if (hipMalloc(reinterpret_cast<void**>(&d.d_flags2), 2 * sizeof(int)) != hipSuccess) return RPM_E_DEVICE;
    if (hipHostMalloc(reinterpret_cast<void**>(&d.h_flags2), 2 * sizeof(int), hipHostMallocDefault) != hipSuccess) return RPM_E_DEVICE;
    d.h_flags2[0] = d.h_flags2[1] = 0;
  }
  HIP_TRY(e, hipMemsetAsync(d.d_flags2 + slot, 0, sizeof(int), d.stream));
  unsigned blocks = unsigned((count + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(rpm_finite_kernel, dim3(blocks), dim3(256), 0, d.stream, dev, count, d.d_flags2 + slot);
  return RPM_OK;
}
int dev_flags_fetch(Engine& e) {
  Device& d = *e.dev;
  HIP_TRY(e, hipMemcpyAsync(d.h_flags2, d.d_flags2, 2 * sizeof(int), hipMemcpyDeviceToHost, d.stream));
  return RPM_OK;
}
int dev_flag_value(Engine& e, int slot) { return e.dev->h_flags2 ? e.dev->h_flags2[slot] : 0; }
int dev_download_enqueue(Engine& e, double* host, const double* dev, size_t count) {
  HIP_TRY(e, hipMemcpyAsync(host, dev, count * sizeof(double), hipMemcpyDeviceToHost, e.dev->stream));
  return RPM_OK;
}

// Page-lock the caller's buffer once (Ipopt hands the same x / g / values arrays every iteration) so that
// the copies are direct DMA at PCIe rate instead of staged pageable copies.  Best effort: failures are ignored.
void dev_pin_host(Engine& e, const void* ptr, size_t bytes) {
  if (!e.opt_pin_host || !ptr || bytes < (64u << 10)) return;
  Device& d = *e.dev;
  for (auto& p : d.pinned)
    if (p.first == ptr && p.second >= bytes) return;
  for (auto it = d.pinned.begin(); it != d.pinned.end(); ++it)
    if (it->first == ptr) {   // same address, grew: re-register
      (void)hipHostUnregister(const_cast<void*>(it->first));
      (void)hipGetLastError();
      d.pinned.erase(it);
      break;
    }
  if (hipHostRegister(const_cast<void*>(ptr), bytes, hipHostRegisterDefault) == hipSuccess)
    d.pinned.emplace_back(ptr, bytes);
  else
    (void)hipGetLastError();
}

// ---- small helpers used by the C ABI (rpm_abi.cpp) ---------------------------------------------
int dev_upload_x(Engine& e, const double* x) {
  if (!e.dev) {
    int rc = device_init(e, 0);
    if (rc) return rc;
  }
  HIP_TRY(e, hipSetDevice(e.dev->device_id));
  HIP_TRY(e, hipMemcpyAsync(e.dev->d_x, x, size_t(e.n_instances) * e.n * sizeof(double), hipMemcpyHostToDevice,
                            e.dev->stream));
  return RPM_OK;
}
int dev_download(Engine& e, double* host, const double* dev, size_t count) {
  HIP_TRY(e, hipMemcpyAsync(host, dev, count * sizeof(double), hipMemcpyDeviceToHost, e.dev->stream));
  HIP_TRY(e, hipStreamSynchronize(e.dev->stream));
  return RPM_OK;
}
int dev_upload(Engine& e, double* dev, const double* host, size_t count) {
  HIP_TRY(e, hipMemcpyAsync(dev, host, count * sizeof(double), hipMemcpyHostToDevice, e.dev->stream));
  return RPM_OK;
}
int dev_sync(Engine& e) {
  if (!e.dev) return RPM_OK;
  HIP_TRY(e, hipStreamSynchronize(e.dev->stream));
  return RPM_OK;
}
double* dev_buf(Engine& e, int which) {
  Device& d = *e.dev;
  switch (which) {
    case 0: return d.d_x;
    case 1: return d.d_g;
    case 2: return d.d_values;
    case 3: return d.d_grad;
    case 4: return d.d_obj;
    case 5: return d.d_lambda;
    case 6: return d.d_hess;
  }
  return nullptr;
}
bool& dev_cache_valid(Engine& e) { return e.dev->cache_valid; }
void* dev_stream(Engine& e) { return e.dev->stream; }

// ---- interval sharding: pack a rank's runs / scatter the gathered runs of every rank -----------
struct SegCopy { int src, dst, len, pad; };
__global__ void rpm_seg_copy_kernel(const SegCopy* __restrict__ segs, const double* __restrict__ src,
                                    double* __restrict__ dst) {
  const SegCopy s = segs[blockIdx.x];
  for (int i = blockIdx.y * blockDim.x + threadIdx.x; i < s.len; i += gridDim.y * blockDim.x)
    dst[s.dst + i] = src[s.src + i];
}

int dev_shard_copy(Engine& e, int which, bool pack, const double* src, int stride, double* dst, void* stream) {
  if (!e.dev) {
    int rc = device_init(e, 0);
    if (rc) return rc;
  }
  Device& d = *e.dev;
  hipStream_t st = static_cast<hipStream_t>(stream);
  Device::SegTable& tab = d.segtab[which][pack ? 0 : 1];
  if (!tab.ptr || tab.stride != stride) {   // built once per (vector, direction, stride)
    std::vector<SegCopy> segs;
    if (pack) {
      for (const rpm_segment& s : shard_segments(e, which, e.shard_rank, nullptr)) segs.push_back(SegCopy{s.off, s.pos, s.len, 0});
    } else {
      for (int r = 0; r < e.shard_world; ++r)
        for (const rpm_segment& s : shard_segments(e, which, r, nullptr)) segs.push_back(SegCopy{r * stride + s.pos, s.off, s.len, 0});
    }
    if (tab.ptr) HIP_TRY(e, hipFree(tab.ptr));
    tab.ptr = nullptr;
    tab.count = int(segs.size());
    tab.stride = stride;
    if (tab.count) {
      HIP_TRY(e, hipMalloc(&tab.ptr, segs.size() * sizeof(SegCopy)));
      HIP_TRY(e, hipMemcpy(tab.ptr, segs.data(), segs.size() * sizeof(SegCopy), hipMemcpyHostToDevice));
    }
  }
  if (!tab.count) return RPM_OK;
  hipLaunchKernelGGL(rpm_seg_copy_kernel, dim3(unsigned(tab.count), 4), dim3(256), 0, st,
                     static_cast<const SegCopy*>(tab.ptr), src, dst);
  HIP_TRY(e, hipGetLastError());
  return RPM_OK;
}

// ---- exact-Hessian mode: dependency probe (once per mesh) and evaluation ---------------------------
int ensure_hessian(Engine& e) {
  if (e.hess_ready && e.dev && e.dev->d_hpairs) return RPM_OK;
  if (!e.dev) {
    int rc = device_init(e, 0);
    if (rc) return rc;
  }
  Device& d = *e.dev;
  HIP_TRY(e, hipSetDevice(d.device_id));
  ProblemDims pd;
  problem_dims(e.problem_id, &pd);
  const int nv = pd.nx + pd.nu, nout = pd.nx + pd.nc;
  if (!e.hess_ready) {
    // NaN-propagation probe of the dynamics at node 1 of the guess (LpDerivDependciesChecker.cpp:60-93)
    std::vector<int> off(e.P), dep(size_t(e.P) * nv * nout, 0);
    for (int i = 0; i < e.P; ++i) off[i] = i * nv * nout;
    int *d_dep = nullptr, *d_off = nullptr;
    double* d_guess = nullptr;
    HIP_TRY(e, upload(&d_dep, dep));
    HIP_TRY(e, upload(&d_off, off));
    HIP_TRY(e, upload(&d_guess, e.guess));
    hipError_t s = hipSuccess;
    with_problem(e.problem_id, [&](auto prob) {
      using P = decltype(prob);
      hipLaunchKernelGGL((rpm_dep_probe_kernel<P>), dim3(unsigned(e.P)), dim3(64), 0, d.stream, d.kp, d_guess, d_dep, d_off);
      s = hipGetLastError();
    });
    HIP_TRY(e, s);
    HIP_TRY(e, hipStreamSynchronize(d.stream));
    HIP_TRY(e, hipMemcpy(dep.data(), d_dep, dep.size() * sizeof(int), hipMemcpyDeviceToHost));
    (void)hipFree(d_dep);
    (void)hipFree(d_off);
    (void)hipFree(d_guess);
    e.hess_dep.assign(e.P, {});
    for (int i = 0; i < e.P; ++i) e.hess_dep[i].assign(dep.begin() + off[i], dep.begin() + off[i] + nv * nout);
    build_hessian_tables(e);
  }
  // tiles of the Hessian kernel: TH nodes x NR roles per workgroup
  const int NV = nv + 1, NR = (NV + 1) * (NV + 2) / 2, NF = nout + 1;
  int TH = 64;
  while (TH > 1 && TH * NR > 1024) TH /= 2;
  if (TH * NR > 1024) {
    e.err = "exact Hessian: too many variables per node for one workgroup";
    return RPM_E_UNSUPPORTED;
  }
  std::vector<int> tiles;
  for (int ip = 0; ip < e.P; ++ip)
    for (int k0 = 0; k0 < e.ph[ip].N; k0 += TH) {
      tiles.push_back(ip);
      tiles.push_back(k0);
      tiles.push_back(e.ph[ip].N - k0 < TH ? e.ph[ip].N - k0 : TH);
    }
  HIP_TRY(e, upload(&d.d_hpairs, e.hess_pairs));
  HIP_TRY(e, upload(&d.d_hphases, e.hess_phases));
  HIP_TRY(e, upload(&d.d_hends, e.hess_ends));
  HIP_TRY(e, upload(&d.d_hlinks, e.hess_links));
  HIP_TRY(e, upload(&d.d_htiles, tiles));
  const size_t B = size_t(e.n_instances);
  HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&d.d_htmp), B * (e.hess_tmp_len ? e.hess_tmp_len : 1) * sizeof(double)));
  HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&d.d_hess), B * (e.nnz_h ? e.nnz_h : 1) * sizeof(double)));
  d.hp.pairs = d.d_hpairs;
  d.hp.phases = d.d_hphases;
  d.hp.ends = d.d_hends;
  d.hp.links = d.d_hlinks;
  d.hp.tiles = d.d_htiles;
  d.hp.n_tiles = int(tiles.size() / 3);
  d.hp.th = TH;
  d.hp.n_ends = int(e.hess_ends.size());
  d.hp.n_links = int(e.hess_links.size());
  d.hp.nnz_h = e.nnz_h;
  d.hp.tmp_len = e.hess_tmp_len;
  d.hess_threads = ((TH * NR + 63) / 64) * 64;
  d.hess_lds = size_t(NR) * NF * TH * sizeof(double);
  return RPM_OK;
}

int dev_eval_h(Engine& e, const double* d_x, double obj_factor, const double* d_lambda, double* d_values, void* stream) {
  int rc = ensure_hessian(e);
  if (rc) return rc;
  Device& d = *e.dev;
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bool an = e.first_derive == RPM_DERIVE_ANALYTIC;
  hipError_t s = hipSuccess;
  with_problem(e.problem_id, [&](auto prob) {
    using P = decltype(prob);
    dim3 grid(unsigned(d.hp.n_tiles), unsigned(e.n_instances));
    auto launch = [&](auto kern) {
      if (d.hess_lds > 64 * 1024)
        s = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(d.hess_lds));
      if (s == hipSuccess) {
        hipLaunchKernelGGL(kern, grid, dim3(unsigned(d.hess_threads)), d.hess_lds, st, d.kp, d.hp, d_x, obj_factor, d_lambda,
                           d_values, d.d_htmp);
        s = hipGetLastError();
      }
    };
    bool done = false;
    if constexpr (P::HAS_ANALYTIC) {
      if (an) {
        launch(rpm_hess_kernel<P, true>);
        done = true;
      }
    }
    if (!done) launch(rpm_hess_kernel<P, false>);
    if (s != hipSuccess) return;
    hipLaunchKernelGGL(rpm_hess_tt_kernel, dim3(unsigned(e.P), unsigned(e.n_instances)), dim3(256), 0, st, d.kp, d.hp,
                       d.d_htmp, d_values);
    const int ne = d.hp.n_ends + d.hp.n_links;
    if (ne > 0)
      hipLaunchKernelGGL((rpm_hess_end_kernel<P>), dim3(unsigned((ne + 127) / 128), unsigned(e.n_instances)), dim3(128), 0, st,
                         d.kp, d.hp, d_x, obj_factor, d_lambda, d_values);
    s = hipGetLastError();
  });
  if (s != hipSuccess) {
    e.err = std::string("rpm_hess_kernel launch: ") + hipGetErrorString(s);
    return RPM_E_DEVICE;
  }
  return RPM_OK;
}

// Nlp2OpControl for one phase: host x / lambda in, (N+1)-row column-major host arrays out (any may be NULL)
int dev_nlp2op(Engine& e, int phase, const double* x, const double* lambda, double* time, double* state, double* control,
               double* costate, double* pathmult, double* hamiltonian, double* mayer_cost, double* lagrange_cost) {
  if (!e.dev) {
    int rc = device_init(e, 0);
    if (rc) return rc;
  }
  Device& d = *e.dev;
  HIP_TRY(e, hipSetDevice(d.device_id));
  const PhaseHost& p = e.ph[phase];
  const int N = p.N, M = N + 1, nx = p.nx, nu = p.nu, nc = p.nc;
  const size_t out_doubles = size_t(M) * (3 + 2 * nx + nu + nc) + 8 + nu + nc;
  double* buf = nullptr;
  HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&buf), out_doubles * sizeof(double)));
  double* o_time = buf;
  double* o_state = o_time + M;
  double* o_control = o_state + size_t(M) * nx;
  double* o_costate = o_control + size_t(M) * nu;
  double* o_pathmult = o_costate + size_t(M) * nx;
  double* o_ham = o_pathmult + size_t(M) * nc;
  double* o_lag = o_ham + M;
  double* o_scal = o_lag + M;          // [0] mayer, [1] lagrange cost
  double* u_end = o_scal + 8;
  double* pm_end = u_end + nu;
  int rc = dev_upload(e, d.d_x, x, size_t(e.n));
  if (rc == RPM_OK) rc = dev_upload(e, d.d_lambda, lambda, size_t(e.m));
  hipError_t s = hipSuccess;
  if (rc == RPM_OK) {
    hipStream_t st = d.stream;
    const PhaseDev& q = e.phd[phase];
    const double tspan = x[q.x_t0 + 1] - x[q.x_t0];
    if (nu > 0)
      hipLaunchKernelGGL(rpm_post_spline_kernel, dim3(1), dim3(64), 0, st, N, d.d_points + q.node0, d.d_x + q.x_control0, nu,
                         1.0, 1.0, static_cast<const double*>(nullptr), u_end);
    if (nc > 0)
      hipLaunchKernelGGL(rpm_post_spline_kernel, dim3(1), dim3(64), 0, st, N, d.d_points + q.node0,
                         d.d_lambda + size_t(N) * nx, nc, 2.0, tspan, d.d_weights + q.node0, pm_end);
    with_problem(e.problem_id, [&](auto prob) {
      using P = decltype(prob);
      hipLaunchKernelGGL((rpm_post_kernel<P>), dim3(unsigned((M + 255) / 256)), dim3(256), 0, st, d.kp, phase, d.d_x, d.d_lambda,
                         u_end, pm_end, o_time, o_state, o_control, o_costate, o_pathmult, o_ham, o_lag, o_scal);
    });
    hipLaunchKernelGGL(rpm_post_cost_kernel, dim3(1), dim3(256), 0, st, d.kp, phase, d.d_x, o_lag, o_scal + 1);
    s = hipGetLastError();
    if (s == hipSuccess) s = hipStreamSynchronize(st);
    auto get = [&](double* host, const double* dev, size_t cnt) {
      if (host && cnt && s == hipSuccess) s = hipMemcpy(host, dev, cnt * sizeof(double), hipMemcpyDeviceToHost);
    };
    get(time, o_time, M);
    get(state, o_state, size_t(M) * nx);
    get(control, o_control, size_t(M) * nu);
    get(costate, o_costate, size_t(M) * nx);
    get(pathmult, o_pathmult, size_t(M) * nc);
    get(hamiltonian, o_ham, M);
    get(mayer_cost, o_scal, 1);
    get(lagrange_cost, o_scal + 1, 1);
  }
  (void)hipFree(buf);
  if (rc) return rc;
  if (s != hipSuccess) {
    e.err = std::string("nlp2op: ") + hipGetErrorString(s);
    return RPM_E_DEVICE;
  }
  return RPM_OK;
}

// CheckSolutionDiffError for one phase: host x in, relative_error ((N + K + 1) x nx, column-major) out
int dev_solution_error(Engine& e, int phase, const double* x, double* rel_err) {
  if (!e.dev) {
    int rc = device_init(e, 0);
    if (rc) return rc;
  }
  Device& d = *e.dev;
  HIP_TRY(e, hipSetDevice(d.device_id));
  const PhaseHost& p = e.ph[phase];
  if (e.mesh_err.size() != e.ph.size()) e.mesh_err.assign(e.ph.size(), MeshErrTables());
  MeshErrTables& t = e.mesh_err[phase];
  if (t.iv.empty()) build_mesh_err_tables(p, t);
  const int rows = t.rows, nx = p.nx, nu = p.nu, K = int(t.iv.size());
  int nmax = 0;
  for (const MeshIvDev& iv : t.iv) nmax = std::max(nmax, iv.n + 1);
  const size_t lds = sizeof(double) * size_t(nmax) * (2 * nx + nu);
  if (lds > 60 * 1024) {
    e.err = "solution_error: a mesh interval has too many nodes for the estimator's LDS tile";
    return RPM_E_UNSUPPORTED;
  }
  // one device block: doubles first, then the ints
  const size_t nd = t.ttem.size() + t.Hs.size() + t.Ss.size() + t.Hc.size() + t.Sc.size() + t.A.size() + 3 * size_t(rows) * nx;
  const size_t ni = t.hit_s.size() + t.hit_c.size();
  const size_t bytes = nd * sizeof(double) + ni * sizeof(int) + K * sizeof(MeshIvDev);
  char* buf = nullptr;
  HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&buf), bytes));
  double* dd = reinterpret_cast<double*>(buf);
  double* d_ttem = dd; dd += t.ttem.size();
  double* d_Hs = dd; dd += t.Hs.size();
  double* d_Ss = dd; dd += t.Ss.size();
  double* d_Hc = dd; dd += t.Hc.size();
  double* d_Sc = dd; dd += t.Sc.size();
  double* d_A = dd; dd += t.A.size();
  double* d_fine = dd; dd += size_t(rows) * nx;
  double* d_integ = dd; dd += size_t(rows) * nx;
  double* d_rel = dd; dd += size_t(rows) * nx;
  int* d_hit_s = reinterpret_cast<int*>(dd);
  int* d_hit_c = d_hit_s + t.hit_s.size();
  MeshIvDev* d_iv = reinterpret_cast<MeshIvDev*>(d_hit_c + t.hit_c.size());
  hipError_t s = hipSuccess;
  auto put = [&](void* dev, const void* host, size_t cnt) {
    if (cnt && s == hipSuccess) s = hipMemcpy(dev, host, cnt, hipMemcpyHostToDevice);
  };
  put(d_ttem, t.ttem.data(), t.ttem.size() * sizeof(double));
  put(d_Hs, t.Hs.data(), t.Hs.size() * sizeof(double));
  put(d_Ss, t.Ss.data(), t.Ss.size() * sizeof(double));
  put(d_Hc, t.Hc.data(), t.Hc.size() * sizeof(double));
  put(d_Sc, t.Sc.data(), t.Sc.size() * sizeof(double));
  put(d_A, t.A.data(), t.A.size() * sizeof(double));
  put(d_hit_s, t.hit_s.data(), t.hit_s.size() * sizeof(int));
  put(d_hit_c, t.hit_c.data(), t.hit_c.size() * sizeof(int));
  put(d_iv, t.iv.data(), K * sizeof(MeshIvDev));
  int rc = (s == hipSuccess) ? dev_upload(e, d.d_x, x, size_t(e.n)) : RPM_OK;
  if (rc == RPM_OK && s == hipSuccess) {
    hipStream_t st = d.stream;
    with_problem(e.problem_id, [&](auto prob) {
      using P = decltype(prob);
      hipLaunchKernelGGL((rpm_mesh_err_kernel<P>), dim3(unsigned(K)), dim3(128), lds, st, d.kp, phase, d.d_x, d_iv, K, d_Hs,
                         d_Ss, d_hit_s, d_Hc, d_Sc, d_hit_c, d_A, d_ttem, rows, d_fine, d_integ);
    });
    hipLaunchKernelGGL(rpm_mesh_rel_kernel, dim3(unsigned(nx)), dim3(256), 0, st, rows, d_fine, d_integ, d_rel);
    s = hipGetLastError();
    if (s == hipSuccess) s = hipStreamSynchronize(st);
    if (s == hipSuccess) s = hipMemcpy(rel_err, d_rel, size_t(rows) * nx * sizeof(double), hipMemcpyDeviceToHost);
  }
  (void)hipFree(buf);
  if (rc) return rc;
  if (s != hipSuccess) {
    e.err = std::string("solution_error: ") + hipGetErrorString(s);
    return RPM_E_DEVICE;
  }
  return RPM_OK;
}

}  // namespace rpm

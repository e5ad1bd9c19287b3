// rpm_hess_kernels.hip — K6, the exact Lagrangian Hessian (hessian-approximation=exact): NaN-propagation dependency probe,
// second-difference kernels, their host drivers.  Reference: Core/LpHessian.cpp, Core/LpDerivDependciesChecker.cpp.
#include "rpm_device_internal.hpp"

namespace rpm {

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
  pf_dae<Prob>(ph.phase_num, tk, xs, us, xg + ph.x_t0 + 2, K.consts, f, cp);
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
  const double* c = K.consts + size_t(inst) * K.consts_stride;
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
  if constexpr (has_stage<Prob>::value) {
    // functors with staged dynamics (problems.hpp): role 0 of a node leaves the sub-expressions of the unperturbed point in LDS,
    // every perturbed evaluation recomputes only what its one or two variables enter — same operations, same bits
    using Stage = typename stage_of<Prob>::type;
    constexpr int SD = int(sizeof(Stage) / sizeof(double));
    __shared__ double stg[SD * 64];
    if (role == 0) {             // (unperturbed: a = b = -1)
      Stage s0;
      Prob::stage(ph.phase_num, tk, xs, us, c, s0);
#pragma unroll
      for (int i = 0; i < SD; ++i) stg[i * 64 + kk] = reinterpret_cast<const double*>(&s0)[i];
    }
    __syncthreads();
    Stage sb;
#pragma unroll
    for (int i = 0; i < SD; ++i) reinterpret_cast<double*>(&sb)[i] = stg[i * 64 + kk];
    double cp[NCs];
    Prob::dae_from2(ph.phase_num, tk, xs, us, c, sb, a, b, F, cp);
#pragma unroll
    for (int j = 0; j < NC; ++j) F[NX + j] = cp[j];
    F[NX + NC] = pf_lagrange<Prob>(ph.phase_num, tk, xs, us, x + ph.x_t0 + 2, c);
  } else {
    double cp[NCs];
    pf_dae<Prob>(ph.phase_num, tk, xs, us, x + ph.x_t0 + 2, c, F, cp);
#pragma unroll
    for (int j = 0; j < NC; ++j) F[NX + j] = cp[j];
    F[NX + NC] = pf_lagrange<Prob>(ph.phase_num, tk, xs, us, x + ph.x_t0 + 2, c);
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
      pf_dae_jac_col<Prob>(ph.phase_num, b, tk0, xs0, us0, x + ph.x_t0 + 2, c, df, dc);
#pragma unroll
      for (int o = 0; o < NX; ++o) {
        const double term = lam[o * N + k] * df[o];
        sdd = (o == 0) ? term : sdd + term;
      }
      dL = pf_lagrange_grad_col<Prob>(ph.phase_num, b, tk0, xs0, us0, x + ph.x_t0 + 2, c);
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
  const double* c = K.consts + size_t(inst) * K.consts_stride;
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
      if (ph.ne > 0) pf_event<Prob>(ph.phase_num, s0, y0, sf, yf, x + ph.x_t0 + 2, c, ev[q]);
      my[q] = pf_mayer<Prob>(ph.phase_num, s0, y0, sf, yf, x + ph.x_t0 + 2, c);
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
      pf_link<Prob>(lk.left + 1, lk.right + 1, w, w + NX, x + K.phases[lk.left].x_t0 + 2, x + K.phases[lk.right].x_t0 + 2, c, lk.nlink, lo[q]);
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

}  // namespace rpm

// rpm_device.hip — device half of the engine that is not a kernel family of its own: initialisation and teardown, the
// objective kernel (K5: objective quadrature and gradient, GetObjFun LpNLPWrapper.cpp:863-939, GetObjGrad :940-1104),
// transfers and NaN/Inf scans of the host-pointer path, the interval-sharding copies.  The tile kernels (K1-K4) live in
// rpm_tile_kernels.hip, the exact Hessian (K6) in rpm_hess_kernels.hip, the post-solve kernels in rpm_post_kernels.hip.
// No CPU fallback lives here or anywhere else in the product.
#include "rpm_device_internal.hpp"
#include "rpm_pin.h"

namespace rpm {

// ------------------------------------------------------------------------------------------
// Objective and gradient.  One workgroup per phase; thread = node (strided).  Sums use a fixed
// binary tree over the workgroup so the result is deterministic (independent of timing).
template <class Prob, bool GRAD, bool AN>
__global__ void rpm_obj_kernel(const KParams K, const double* __restrict__ xall, double* __restrict__ objall,
                               double* __restrict__ gradall, double* __restrict__ partial) {
  constexpr int NX = Prob::NX, NU = Prob::NU, NQ = prob_nq<Prob>::value;
  constexpr int NXs = NX > 0 ? NX : 1, NUs = NU > 0 ? NU : 1, NQs = NQ > 0 ? NQ : 1;
  __shared__ double red[2 + NQs][256];
  const int tid = threadIdx.x;
  const int p = blockIdx.x;
  const int inst = blockIdx.y;
  const double* __restrict__ x = xall + size_t(inst) * K.n;
  double* grad = GRAD ? gradall + size_t(inst) * K.n : nullptr;
  const PhaseDev ph = K.phases[p];
  const double* c = K.consts + size_t(inst) * K.consts_stride;
  const int N = ph.N;
  bool bad = false;   // NaN/Inf among the gradient entries this thread stores (host-pointer path, K.chk)
  const double t0 = x[ph.x_t0], tf = x[ph.x_t0 + 1];
  const double tspan = tf - t0;
  double s_wl = 0.0, s_t0 = 0.0;   // sum w_k L_k ; sum (w_k dt/2 dL/dt)_k (1-tau_k)/2
  double s_p[NQs];                 // sum (w_k dt/2) dL/dp_j  (LpNLPWrapper.cpp:1088-1097: the quadrature of the parameter column)
  double pq[NQs];                  // the phase's static parameters
#pragma unroll
  for (int j = 0; j < NQs; ++j) { s_p[j] = 0.0; pq[j] = j < NQ ? x[ph.x_t0 + 2 + j] : 0.0; }
  for (int k = tid; k < N; k += blockDim.x) {
    const double tau = K.points[ph.node0 + k], w = K.weights[ph.node0 + k];
    const double tk = (tau + 1) * (tspan / 2.0) + t0;
    double xs[NXs], us[NUs];
#pragma unroll
    for (int i = 0; i < NX; ++i) xs[i] = x[ph.x_state0 + i * (N + 1) + k];
#pragma unroll
    for (int j = 0; j < NU; ++j) us[j] = x[ph.x_control0 + j * N + k];
    const double L0 = pf_lagrange<Prob>(ph.phase_num, tk, xs, us, pq, c);
    s_wl += w * L0;
    if (GRAD) {
      const double wk = w * tspan / 2.0;                       // Weights*tspan/2.0, :1051
      double dLt;
      if constexpr (AN) {
#pragma unroll
        for (int i = 0; i < NX; ++i)
          { const double gv_ = wk * pf_lagrange_grad_col<Prob>(ph.phase_num, i, tk, xs, us, pq, c); grad[ph.x_state0 + i * (N + 1) + k] = gv_; chk_note(bad, gv_); }
#pragma unroll
        for (int j = 0; j < NU; ++j)
          { const double gv_ = wk * pf_lagrange_grad_col<Prob>(ph.phase_num, NX + j, tk, xs, us, pq, c); grad[ph.x_control0 + j * N + k] = gv_; chk_note(bad, gv_); }
        dLt = pf_lagrange_grad_col<Prob>(ph.phase_num, NX + NU, tk, xs, us, pq, c);
#pragma unroll
        for (int j = 0; j < NQ; ++j) s_p[j] += wk * pf_lagrange_grad_col<Prob>(ph.phase_num, NX + NU + 1 + j, tk, xs, us, pq, c);
      } else {
        // LpFDderive::DerivLagrange, LpFiniteDifferenceDerive.cpp:100-192
#pragma unroll
        for (int i = 0; i < NX; ++i) {
          const double b = xs[i], hh = K.tol * (1 + fabs(b));
          xs[i] = b + hh;
          const double Lp = pf_lagrange<Prob>(ph.phase_num, tk, xs, us, pq, c);
          xs[i] = b;
          { const double gv_ = wk * ((Lp - L0) / hh); grad[ph.x_state0 + i * (N + 1) + k] = gv_; chk_note(bad, gv_); }
        }
#pragma unroll
        for (int j = 0; j < NU; ++j) {
          const double b = us[j], hh = K.tol * (1 + fabs(b));
          us[j] = b + hh;
          const double Lp = pf_lagrange<Prob>(ph.phase_num, tk, xs, us, pq, c);
          us[j] = b;
          { const double gv_ = wk * ((Lp - L0) / hh); grad[ph.x_control0 + j * N + k] = gv_; chk_note(bad, gv_); }
        }
        const double ht = K.tol * (1 + fabs(tk));
        dLt = (pf_lagrange<Prob>(ph.phase_num, tk + ht, xs, us, pq, c) - L0) / ht;
#pragma unroll
        for (int j = 0; j < NQ; ++j) {   // LpFDderive::DerivLagrange's parameter columns, LpFiniteDifferenceDerive.cpp:165-182
          const double b = pq[j], hh = K.tol * (1 + fabs(b));
          pq[j] = b + hh;
          const double Lp = pf_lagrange<Prob>(ph.phase_num, tk, xs, us, pq, c);
          pq[j] = b;
          s_p[j] += wk * ((Lp - L0) / hh);
        }
      }
      s_t0 += ((w * (tspan / 2.0)) * dLt) * (tau * (-0.5) + 0.5);   // ret2*ret3, :1072-1077
    }
  }
  red[0][tid] = s_wl;
  red[1][tid] = s_t0;
#pragma unroll
  for (int j = 0; j < NQ; ++j) red[2 + j][tid] = s_p[j];
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if (tid < s) {
      red[0][tid] += red[0][tid + s];
      red[1][tid] += red[1][tid + s];
#pragma unroll
      for (int j = 0; j < NQ; ++j) red[2 + j][tid] += red[2 + j][tid + s];
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
    const double mayer = pf_mayer<Prob>(ph.phase_num, t0, x0, tf, xf, pq, c);
    // per-phase cost  Mayer + (w'L)(dt/2)  (:926-932); phases are summed in order by phase 0's thread below
    partial[size_t(inst) * K.P + p] = mayer + wl * (tspan / 2.0);
    if (GRAD) {
      // Mayer derivative w.r.t. [x0.., t0, xf.., tf]  (LpFDderive::DerivMayer :11-98 or the analytic callback)
      double dM[2 * NXs + 2 + NQs];   // [x0.., t0, xf.., tf, p..]  (LpFDderive::DerivMayer)
      if constexpr (AN) {
        for (int q = 0; q < 2 * NX + 2 + NQ; ++q) dM[q] = pf_mayer_grad_col<Prob>(ph.phase_num, q, t0, x0, tf, xf, pq, c);
      } else {
        const double m0 = mayer;
        const double h0 = K.tol * (1 + fabs(t0)), hf = K.tol * (1 + fabs(tf));
        dM[NX] = (pf_mayer<Prob>(ph.phase_num, t0 + h0, x0, tf, xf, pq, c) - m0) / h0;
        dM[2 * NX + 1] = (pf_mayer<Prob>(ph.phase_num, t0, x0, tf + hf, xf, pq, c) - m0) / hf;
        for (int j = 0; j < NQ; ++j) {
          const double b = pq[j], hb = K.tol * (1 + fabs(b));
          pq[j] = b + hb;
          dM[2 * NX + 2 + j] = (pf_mayer<Prob>(ph.phase_num, t0, x0, tf, xf, pq, c) - m0) / hb;
          pq[j] = b;
        }
        for (int i = 0; i < NX; ++i) {
          const double b0 = x0[i], hb0 = K.tol * (1 + fabs(b0));
          x0[i] = b0 + hb0;
          dM[i] = (pf_mayer<Prob>(ph.phase_num, t0, x0, tf, xf, pq, c) - m0) / hb0;
          x0[i] = b0;
          const double bf = xf[i], hbf = K.tol * (1 + fabs(bf));
          xf[i] = bf + hbf;
          dM[NX + 1 + i] = (pf_mayer<Prob>(ph.phase_num, t0, x0, tf, xf, pq, c) - m0) / hbf;
          xf[i] = bf;
        }
      }
      // terminal-state entries (:1054).  The initial-state Mayer entry is overwritten by the Lagrange
      // run in the reference (:1050-1053) — kept, it is zero in every supported problem anyway.
      for (int i = 0; i < NX; ++i) { const double gv_ = dM[NX + 1 + i]; grad[ph.x_state0 + i * (N + 1) + N] = gv_; chk_note(bad, gv_); }
      // d/dt0 (:1069-1078) and d/dtf (:1081-1087, which keeps only node 0 of the dL/dt term)
      { const double gv_ = (red[1][0] + dM[NX]) + (-0.5) * wl; grad[ph.x_t0] = gv_; chk_note(bad, gv_); }
      double dLt0;
      {
        double xs[NXs], us[NUs];
        for (int i = 0; i < NX; ++i) xs[i] = x0[i];
        for (int j = 0; j < NU; ++j) us[j] = x[ph.x_control0 + j * N];
        const double tau = K.points[ph.node0];
        const double tk = (tau + 1) * (tspan / 2.0) + t0;
        if constexpr (AN) {
          dLt0 = pf_lagrange_grad_col<Prob>(ph.phase_num, NX + NU, tk, xs, us, pq, c);
        } else {
          const double ht = K.tol * (1 + fabs(tk));
          dLt0 = (pf_lagrange<Prob>(ph.phase_num, tk + ht, xs, us, pq, c) - pf_lagrange<Prob>(ph.phase_num, tk, xs, us, pq, c)) / ht;
        }
        const double r2 = (K.weights[ph.node0] * (tspan / 2.0)) * dLt0;
        { const double gv_ = (dM[2 * NX + 1] + 0.5 * wl) + (tau * 0.5 + 0.5) * r2; grad[ph.x_t0 + 1] = gv_; chk_note(bad, gv_); }
      }
      // d/dp_j = dMayer/dp_j + sum_k w_k (dt/2) dL/dp_j  (:1088-1097; the reference multiplies two column vectors there and
      // never fills SolCost.parameter_ — SURVEY B-21 — so this is the formula it means, not the code it has)
      for (int j = 0; j < NQ; ++j) { const double gv_ = dM[2 * NX + 2 + j] + red[2 + j][0]; grad[ph.x_t0 + 2 + j] = gv_; chk_note(bad, gv_); }
    }
  }
  (void)objall;
  if (GRAD && K.chk != nullptr && bad) atomicOr_system(K.chk + 3, 1);
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
bool problem_dims(int id, ProblemDims* out) {
  return with_problem(id, [&](auto prob) {
    using P = decltype(prob);
    *out = ProblemDims{P::NX, P::NU, P::NC, P::NE_MAX, P::NLINK_MAX, P::NCONST, P::HAS_ANALYTIC, prob_nq<P>::value};
  });
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
                  d->d_dvals, d->d_doff, d->d_consts, d->d_inst_consts, d->d_alin_v, d->d_links, d->d_alin_j, d->d_x, d->d_g,
                  d->d_values, d->d_grad, d->d_obj, d->d_lambda, d->d_hess, d->d_partial, d->d_hpairs, d->d_hphases,
                  d->d_hends, d->d_hlinks, d->d_htiles, d->d_htmp};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (d->d_flag) (void)hipFree(d->d_flag);
  if (d->h_flags2) (void)hipHostFree(d->h_flags2);   // d_flags2 is its device alias
  host_path_destroy(d);
  exchange_destroy(d);
  rpm_pin_release_owner(&e);   // this engine's holds; pages another engine still addresses stay registered
  dev_stage_destroy(d);
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
  HIP_TRY(e, hipMemsetAsync(d->d_grad, 0, B * e.n * sizeof(double), d->stream));   // ordered with the kernels that follow on this stream
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
  k.consts_stride = 0;
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
  k.sg = e.m;
  k.sv = e.nnz_jac;
  k.max_cshare = 0;
  for (const TileDev& t : e.tiles) k.max_cshare = t.c_cnt > k.max_cshare ? t.c_cnt : k.max_cshare;
  const bool sharded = e.shard_mode == RPM_SHARD_INTERVALS && e.shard_world > 1;
  k.tasks = d->d_tasks;
  k.n_tasks = (!sharded || e.shard_rank == 0) ? int(e.tasks.size()) : 0;  // rank 0 owns the endpoint rows
  k.skip_const = 0;
  k.diag_mask = 0;
  k.trace = nullptr;
  k.chk = nullptr;
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
  tile_pipeline_setup(e, d, pd, device_id);
  return dev_update_instance_constants(e);
}

int dev_eval_obj(Engine& e, const double* d_x, double* d_obj, double* d_grad, void* stream, bool host_chk) {
  if (!e.dev) {
    int rc = device_init(e, 0);
    if (rc) return rc;
  }
  Device& dd = *e.dev;
  struct { KParams kp; double* d_partial; } d{dd.kp, dd.d_partial};
  d.kp.chk = host_chk ? dd.d_flags2 : nullptr;   // host-pointer path: word 3 receives "a gradient entry is NaN/Inf"
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

__global__ void rpm_finite2_kernel(const double* __restrict__ a, size_t na, const double* __restrict__ b, size_t nb,
                                   int* __restrict__ flags) {
  // flags[0]: a NaN/Inf in a[0..na); flags[1]: in b[0..nb).  The host zeroes both words before the launch.
  bool bad_a = false, bad_b = false;
  const size_t stride = size_t(gridDim.x) * blockDim.x, first = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
  for (size_t i = first; i < na; i += stride) bad_a |= !isfinite(a[i]);
  for (size_t i = first; i < nb; i += stride) bad_b |= !isfinite(b[i]);
  if (__any(bad_a) && (threadIdx.x & 63) == 0) atomicOr_system(flags, 1);       // host memory
  if (__any(bad_b) && (threadIdx.x & 63) == 0) atomicOr_system(flags + 1, 1);
}

// The same check without its own round trip: scan a[0..na) (flag word 0) and b[0..nb) (word 1, may be empty) in ONE
// launch on the engine's stream; the words are device-visible host memory.  The caller synchronises once, together
// with its result download, and reads dev_flag_value.
int dev_nonfinite_enqueue(Engine& e, const double* a, size_t na, const double* b, size_t nb) {
  Device& d = *e.dev;
  if (!d.h_flags2) {   // the two words live in page-locked host memory the device can write: no memset, no copy back
    if (hipHostMalloc(reinterpret_cast<void**>(&d.h_flags2), 4 * sizeof(int), hipHostMallocMapped) != hipSuccess) return RPM_E_DEVICE;
    d.h_flags2[2] = d.h_flags2[3] = 0;
    if (hipHostGetDevicePointer(reinterpret_cast<void**>(&d.d_flags2), d.h_flags2, 0) != hipSuccess) return RPM_E_DEVICE;
  }
  d.h_flags2[0] = d.h_flags2[1] = 0;   // the previous scan was synchronised before its words were read
  const size_t most = na > nb ? na : nb;
  unsigned blocks = unsigned((most + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(rpm_finite2_kernel, dim3(blocks), dim3(256), 0, d.stream, a, na, b, nb, d.d_flags2);
  return RPM_OK;
}
int dev_flag_value(Engine& e, int slot) { return e.dev->h_flags2 ? e.dev->h_flags2[slot] : 0; }

// Page-locked caller arrays (option "pin_host", off unless the caller asks: RpmTNLP does for Ipopt's arrays).  The table
// of registrations is NOT the engine's: hipHostRegister is process-wide, so all engines of all libraries of the process
// share the one registry of librpm_pin.so (rpm_pin.cpp: exactly the arrays' bytes, never a byte twice, reference-counted,
// every refusal counted and kept for rpm_last_error).  An engine holds at most PIN_MAX registrations (least recently used goes
// first) and lets go of all of them in rpm_destroy or when the option is set to 0.  Returns the device-visible alias of
// `ptr`, or nullptr when the array is not page-locked (option off, below RPM_PIN_MIN_BYTES, refused) — the caller of this
// function then goes through the engine's own staging buffers, never through the runtime's pageable-copy path.
constexpr int PIN_MAX = 8;
long dev_pin_counter(int which) { return rpm_pin_counter(which); }
int dev_pin_held(const Engine& e) { return rpm_pin_held(&e); }
void dev_pin_release_all(Engine& e) {
  rpm_pin_release_owner(&e);
  e.pin_refused.clear();
}
std::string dev_pin_last_error() {
  char buf[320];
  buf[0] = 0;
  rpm_pin_last_error(buf, sizeof buf);
  return buf;
}
void* dev_pin_host(Engine& e, const void* ptr, size_t bytes) {
  // an interval-sharded engine stores only ITS rows / runs into the caller's g and values: they have to be addressable
  // whatever their size (a staged copy of the whole array would overwrite the other ranks' shares)
  const bool sharded = e.shard_mode == RPM_SHARD_INTERVALS && e.shard_world > 1;
  if (!e.opt_pin_host || !ptr || bytes == 0 || (bytes < RPM_PIN_MIN_BYTES && !sharded)) return nullptr;
  for (const auto& r : e.pin_refused)   // asked before and refused: staged until the option is set again
    if (r.first == ptr && r.second == bytes) return nullptr;
  const long refused = rpm_pin_counter(RPM_PIN_REGISTER_FAILURES) + rpm_pin_counter(RPM_PIN_OVERLAP_REFUSED);
  void* alias = rpm_pin_acquire(&e, ptr, bytes, PIN_MAX, sharded ? 1 : 0);
  if (!alias && rpm_pin_counter(RPM_PIN_REGISTER_FAILURES) + rpm_pin_counter(RPM_PIN_OVERLAP_REFUSED) != refused)
    e.pin_note = dev_pin_last_error();   // the call goes on through the staging buffers; the reason stays readable
  if (!alias) {
    if (e.pin_refused.size() >= 16) e.pin_refused.erase(e.pin_refused.begin());
    e.pin_refused.emplace_back(ptr, bytes);
  }
  return alias;
}

// ---- the engine's own page-locked staging buffers (hipHostMalloc, mapped) -----------------------------------------------
// Caller arrays that are not registered never reach the HIP runtime: inputs are copied by the CPU into a staging buffer
// the engine owns (the kernels read it in place or a copy engine moves it on), results land in one and are copied out by
// the CPU after the call's synchronisation.  The runtime's pageable-copy path would lock the caller's pages itself and
// keep that lock cached past the call.
int dev_stage_reserve(Engine& e, int slot, size_t count, double** host, double** alias) {
  Device& d = *e.dev;
  Device::Stage& s = d.stage[slot];
  if (s.busy || s.cap < count) {   // a queued copy still reads the slot, or it has to grow
    HIP_TRY(e, hipStreamSynchronize(d.stream));
    dev_stage_synced(e);
  }
  if (s.cap < count) {
    if (s.h) HIP_TRY(e, hipHostFree(s.h));
    s.h = s.d = nullptr;
    s.cap = 0;
    const size_t cap = count + count / 4 + 64;
    HIP_TRY(e, hipHostMalloc(reinterpret_cast<void**>(&s.h), cap * sizeof(double), hipHostMallocMapped));
    HIP_TRY(e, hipHostGetDevicePointer(reinterpret_cast<void**>(&s.d), s.h, 0));
    s.cap = cap;
  }
  *host = s.h;
  if (alias) *alias = s.d;
  return RPM_OK;
}
void dev_stage_synced(Engine& e) {
  if (!e.dev) return;
  for (Device::Stage& s : e.dev->stage) s.busy = false;
}
void dev_stage_destroy(Device* d) {
  for (Device::Stage& s : d->stage) {
    if (s.h) (void)hipHostFree(s.h);
    s = Device::Stage{};
  }
}
// host -> HBM through the staging slot (synchronous with respect to the caller's array: it may be reused on return)
int dev_stage_upload(Engine& e, int slot, double* dev, const double* host, size_t count) {
  if (count == 0) return RPM_OK;
  if (dev_pin_host(e, host, count * sizeof(double))) {   // registered: the copy engine reads the caller's pages
    HIP_TRY(e, hipMemcpyAsync(dev, host, count * sizeof(double), hipMemcpyHostToDevice, e.dev->stream));
    HIP_TRY(e, hipStreamSynchronize(e.dev->stream));     // the caller may reuse its array on return
    dev_stage_synced(e);
    return RPM_OK;
  }
  double* h = nullptr;
  int rc = dev_stage_reserve(e, slot, count, &h, nullptr);
  if (rc) return rc;
  std::memcpy(h, host, count * sizeof(double));
  HIP_TRY(e, hipMemcpyAsync(dev, h, count * sizeof(double), hipMemcpyHostToDevice, e.dev->stream));
  e.dev->stage[slot].busy = true;
  return RPM_OK;
}
// HBM -> host through the staging slot; returns after the data is in the caller's array
int dev_stage_download(Engine& e, int slot, double* host, const double* dev, size_t count) {
  if (count == 0) return RPM_OK;
  if (dev_pin_host(e, host, count * sizeof(double))) {
    HIP_TRY(e, hipMemcpyAsync(host, dev, count * sizeof(double), hipMemcpyDeviceToHost, e.dev->stream));
    HIP_TRY(e, hipStreamSynchronize(e.dev->stream));
    dev_stage_synced(e);
    return RPM_OK;
  }
  double* h = nullptr;
  int rc = dev_stage_reserve(e, slot, count, &h, nullptr);
  if (rc) return rc;
  HIP_TRY(e, hipMemcpyAsync(h, dev, count * sizeof(double), hipMemcpyDeviceToHost, e.dev->stream));
  HIP_TRY(e, hipStreamSynchronize(e.dev->stream));
  dev_stage_synced(e);
  std::memcpy(host, h, count * sizeof(double));
  return RPM_OK;
}

// ---- small helpers used by the C ABI (rpm_abi.cpp) ---------------------------------------------
int dev_upload_x(Engine& e, const double* x) {
  if (!e.dev) {
    int rc = device_init(e, 0);
    if (rc) return rc;
  }
  HIP_TRY(e, hipSetDevice(e.dev->device_id));
  return dev_stage_upload(e, STAGE_X, e.dev->d_x, x, size_t(e.n_instances) * e.n);
}
// `host` is a caller's array: through the page-lock registry or the staging slot, never the runtime's pageable path
int dev_download(Engine& e, double* host, const double* dev, size_t count, int slot) {
  return dev_stage_download(e, slot, host, dev, count);
}
int dev_upload(Engine& e, double* dev, const double* host, size_t count, int slot) {
  return dev_stage_upload(e, slot, dev, host, count);
}
int dev_update_instance_constants(Engine& e) {
  if (!e.dev || e.inst_consts.empty()) return RPM_OK;
  Device& d = *e.dev;
  const size_t count = e.inst_consts.size();
  if (!d.d_inst_consts) HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&d.d_inst_consts), count * sizeof(double)));
  HIP_TRY(e, hipStreamSynchronize(d.stream));
  HIP_TRY(e, hipMemcpy(d.d_inst_consts, e.inst_consts.data(), count * sizeof(double), hipMemcpyHostToDevice));
  d.kp.consts = d.d_inst_consts;
  d.kp.consts_stride = int(e.consts.size());
  host_new_x(e);   // constraint pair, objective and gradient cached under the old constants are gone
  return RPM_OK;
}

int dev_sync(Engine& e) {
  if (!e.dev) return RPM_OK;
  HIP_TRY(e, hipStreamSynchronize(e.dev->stream));
  dev_stage_synced(e);
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
void dev_forget_persistent(Engine& e) {
  if (e.dev) e.dev->const_filled.clear();
}
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

}  // namespace rpm

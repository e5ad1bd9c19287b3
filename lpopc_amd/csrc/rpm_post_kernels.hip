// rpm_post_kernels.hip — the steps after the NLP solve: solution extraction (Nlp2OpConverter) and the mesh-error
// estimate (SolutionErrorChecker), kernels and host drivers.  Once per mesh, not on the metric.
#include "rpm_device_internal.hpp"

namespace rpm {

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
    pf_dae<Prob>(ph.phase_num, t, xs, us, x + ph.x_t0 + 2, K.consts, f, cp);
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
  pf_dae<Prob>(ph.phase_num, t, xs, us, x + ph.x_t0 + 2, K.consts, f, cp);
  const double L = pf_lagrange<Prob>(ph.phase_num, t, xs, us, x + ph.x_t0 + 2, K.consts);
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
    o_mayer[0] = pf_mayer<Prob>(ph.phase_num, t0, x0, tf, xf, x + ph.x_t0 + 2, K.consts);
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
  int rc = dev_upload(e, d.d_x, x, size_t(e.n), STAGE_X);
  if (rc == RPM_OK) rc = dev_upload(e, d.d_lambda, lambda, size_t(e.m), STAGE_LAMBDA);
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
    auto get = [&](double* host, const double* dev, size_t cnt) {   // caller arrays: through the staging slot
      if (host && cnt && s == hipSuccess && rc == RPM_OK) rc = dev_download(e, host, dev, cnt, STAGE_HESS);
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
  int rc = (s == hipSuccess) ? dev_upload(e, d.d_x, x, size_t(e.n), STAGE_X) : RPM_OK;
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
    if (s == hipSuccess) rc = dev_download(e, rel_err, d_rel, size_t(rows) * nx, STAGE_HESS);
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

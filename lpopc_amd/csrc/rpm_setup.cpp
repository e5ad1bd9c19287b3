// rpm_setup.cpp — per-mesh host set-up: sizes, bounds, NLP layout, guess, collocation tables,
// Jacobian structure and the workgroup tiling.  Runs once per mesh, needs no GPU.
//
// Reference counterparts (relative to /root/reference/Lpopc/src):
//   sizes     LpSizeChecker::GetSize              Core/LpSizeChecker.cpp:13-152
//   mesh      MeshRefiner::SetAndCheckMesh        Core/LpMeshRefiner.cpp:10-62
//   bounds    LpBoundsChecker::GetBounds          Core/LpBoundsChecker.cpp:13-348
//   guess     LpGuessChecker::GetGuess            Core/LpGuessChecker.cpp:11-294
//   tables    RPMGenerator::initialize/CollocD/CompositeD/GetLGRPointsImp
//                                                 Core/RPMGenerator.cpp:43-181,253-291
//   structure NLPWrapper::GetPhaseSparsity/GetWholeSparsity/GetConsSparsity
//                                                 Core/LpNLPWrapper.cpp:1106-1578
// Built with -ffp-contract=off so the tables are bit-reproducible (no FMA), as on the
// reference's x86-64 build.
#include <cfloat>
#include <cmath>
#include <cstring>
#include <limits>

#include "rpm_engine.hpp"

namespace rpm {

static const double kPi = 3.14159265358979323846;

// Armadillo 5.300.4 arrayops::accumulate: two interleaved running sums, the order sum(X)
// uses on a column (RPMGenerator.cpp:117 `sum(D)`).
static double accumulate2(const double* a, int n) {
  double s0 = 0.0, s1 = 0.0;
  int j = 1;
  for (; j < n; j += 2) {
    s0 += a[j - 1];
    s1 += a[j];
  }
  if (j - 1 < n) s0 += a[j - 1];
  return s0 + s1;
}

// LGR nodes and weights by Newton iteration on P_{N}(x)+P_{N+1}(x)  (GetLGRPointsImp, :253-291)
void lgr_points(int npts, std::vector<double>& x, std::vector<double>& w) {
  const int N = npts - 1, N1 = npts;
  const double eps = DBL_EPSILON;
  x.assign(N1, 0.0);
  w.assign(N1, 0.0);
  std::vector<double> xold(N1, 2.0);
  std::vector<std::vector<double>> P(N1 + 1, std::vector<double>(N1, 0.0));  // P[col][row]
  for (int k = 0; k <= N; ++k) x[k] = -1 * std::cos(double(k) * ((2 * kPi) / (2 * N + 1)));
  for (int iter = 0; iter < 200; ++iter) {
    double mx = 0.0;
    for (int k = 0; k < N1; ++k) mx = std::fmax(mx, std::fabs(x[k] - xold[k]));
    if (!(mx > eps)) break;
    xold = x;
    for (int r = 0; r < N1; ++r) {
      P[0][r] = 1.0;
      P[1][r] = x[r];
    }
    for (int k = 1; k < N1; ++k)
      for (int r = 0; r < N1; ++r) {
        const double t = x[r] * (2 * k + 1) * P[k][r] - (P[k - 1][r] * k);
        P[k + 1][r] = t / (k + 1);
      }
    for (int r = 1; r <= N; ++r) {
      double t = (1.0 - xold[r]) / N1;
      t = t * (P[N1 - 1][r] + P[N1][r]);
      x[r] = xold[r] - (t / (P[N1 - 1][r] - P[N1][r]));
    }
  }
  w[0] = 2.0 / (N1 * N1);
  for (int r = 1; r <= N; ++r) {
    const double t = P[N][r] * N1;
    w[r] = (1 - x[r]) / (t * t);
  }
}

// Barycentric differentiation matrix on `pts` (N_k LGR nodes + the interval's right end),
// rows of the last point dropped (CollocD, :107-130).  D is (M-1) x M, column-major.
void colloc_d(const std::vector<double>& pts, std::vector<double>& D) {
  const int M = int(pts.size());
  std::vector<double> Y(size_t(M) * M), W(M), B(size_t(M) * M);
  for (int j = 0; j < M; ++j)
    for (int i = 0; i < M; ++i) Y[i + size_t(j) * M] = ((i == j ? 1.0 : 0.0) + pts[i]) - pts[j];
  for (int i = 0; i < M; ++i) {
    double p = 1.0;
    for (int j = 0; j < M; ++j) p *= Y[i + size_t(j) * M];
    W[i] = 1 / p;
  }
  for (int j = 0; j < M; ++j)
    for (int i = 0; i < M; ++i) B[i + size_t(j) * M] = W[i] / (W[j] * Y[i + size_t(j) * M]);
  for (int j = 0; j < M; ++j) B[j + size_t(j) * M] = 1 - accumulate2(&B[size_t(j) * M], M);
  D.assign(size_t(M - 1) * M, 0.0);
  for (int b = 0; b < M; ++b)
    for (int a = 0; a < M - 1; ++a) D[a + size_t(b) * (M - 1)] = -B[b + size_t(a) * M];
}

// RPMGenerator::initialize + CompositeD (:43-105, :132-181)
static void build_tables(PhaseHost& p) {
  p.N = 0;
  for (int v : p.nk) p.N += v;
  p.points.assign(p.N, 0.0);
  p.weights.assign(p.N, 0.0);
  p.nodes.assign(p.N, NodeDev{});
  p.d_i.clear(); p.d_j.clear(); p.d_v.clear();
  p.diag_v.clear();
  p.off_i.clear(); p.off_j.clear(); p.off_v.clear();
  p.drows.clear();
  int row0 = 0;
  for (int s = 0; s < p.K; ++s) {
    const int nk = p.nk[s];
    std::vector<double> x, w, sall(nk + 1), D;
    lgr_points(nk, x, w);
    const double tspan = p.mesh[s + 1] - p.mesh[s];
    for (int r = 0; r < nk; ++r) {
      double v = x[r] + 1;
      v *= tspan / 2.0;
      v += p.mesh[s];
      sall[r] = v;
      p.points[row0 + r] = v;
      double ws = w[r] / 2;
      ws *= tspan;
      p.weights[row0 + r] = ws;
    }
    sall[nk] = p.mesh[s + 1];
    colloc_d(sall, D);
    // COO triplets, column-major sweep of the block (GeneratRowColValue), exact zeros dropped (Find)
    for (int j = 0; j < nk + 1; ++j)
      for (int r = 0; r < nk; ++r) {
        const double d = D[r + size_t(j) * nk];
        const double dd = (r == j) ? d : 0.0;
        const double doff = d - dd;
        if (d != 0.0) {
          p.d_i.push_back(row0 + r);
          p.d_j.push_back(row0 + j);
          p.d_v.push_back(d);
        }
        if (dd != 0.0) p.diag_v.push_back(dd);
        if (doff != 0.0) {
          p.off_i.push_back(row0 + r);
          p.off_j.push_back(row0 + j);
          p.off_v.push_back(doff);
        }
      }
    // device layout: each node's D row stored contiguously (dense per interval, not COO)
    for (int r = 0; r < nk; ++r) {
      NodeDev nd;
      nd.drow_off = int(p.drows.size());
      nd.dcol0 = row0;
      nd.dlen = nk + 1;
      nd.interval = s;
      p.nodes[row0 + r] = nd;
      for (int j = 0; j < nk + 1; ++j) p.drows.push_back(D[r + size_t(j) * nk]);
    }
    row0 += nk;
  }
}

// natural cubic spline of the guess (LpGuessChecker::spline_*, Core/LpGuessChecker.cpp:208-294)
static double spline_eval(double x, const double* xd, const double* yd, int n) {
  std::vector<double> c(n, 0.0), mu(n, 0.0), z(n, 0.0);
  for (int i = 1; i < n - 1; ++i) {
    const double him1 = xd[i] - xd[i - 1], hi = xd[i + 1] - xd[i];
    const double alpha = 3.0 / hi * (yd[i + 1] - yd[i]) - 3.0 / him1 * (yd[i] - yd[i - 1]);
    const double li = 2 * (xd[i + 1] - xd[i - 1]) - him1 * mu[i - 1];
    mu[i] = hi / li;
    z[i] = (alpha - him1 * z[i - 1]) / li;
  }
  c[n - 1] = 0.0;
  for (int j = n - 2; j >= 0; --j) c[j] = z[j] - mu[j] * c[j + 1];
  for (int j = 1; j < n - 1; ++j) c[j] = 2 * c[j];
  int kl = 1, kr = n;
  while (kr - kl > 1) {
    const int k = (kr + kl) / 2;
    if (xd[k - 1] > x) kr = k; else kl = k;
  }
  const double h = xd[kr - 1] - xd[kl - 1];
  const double A = (xd[kr - 1] - x) / h, B = (x - xd[kl - 1]) / h;
  const double Cc = (std::pow(A, 3) - A) * (h * h) / 6.0, Dd = (std::pow(B, 3) - B) * (h * h) / 6.0;
  return A * yd[kl - 1] + B * yd[kr - 1] + Cc * c[kl - 1] + Dd * c[kr - 1];
}

static int bad(Engine& e, int code, const std::string& msg) {
  e.err = msg;
  return code;
}

// Split every phase into runs of at most `T` consecutive nodes, cutting at interval
// boundaries whenever an interval fits (so a tile's D rows and X span stay compact).
void build_tiles(Engine& e, int T) {
  // a workgroup is T nodes x (nx+nu+2) roles and may not exceed 1024 threads
  ProblemDims dims;
  if (!e.role_looped && problem_dims(e.problem_id, &dims))   // the role-looped layout is always T x 4 threads
    while (T > 16 && T * (dims.nx + dims.nu + 2) > 1024) T /= 2;
  e.tile_nodes = T;
  e.tiles.clear();
  e.max_span = e.max_drow = 0;
  for (int ip = 0; ip < e.P; ++ip) {
    PhaseHost& p = e.ph[ip];
    e.phd[ip].tile0 = int(e.tiles.size());
    int k = 0, s = 0, used = 0;  // node cursor, interval cursor, nodes of interval s already tiled
    while (k < p.N) {
      int cnt = 0;
      if (e.role_looped) {
        // throughput layouts: tiles are nodes [T t, T t + T) whatever the interval boundaries (a tile's D rows and X span
        // simply cover the intervals it touches).  Every wave is full, and the 8 T-byte store runs of the Jacobian
        // blocks start on 128-byte lines of each block — a run that straddles lines costs the HBM write path a third of
        // its rate (tools/ubench/store_pattern.py).
        cnt = p.N - k < T ? p.N - k : T;
      }
      // one-role layout: whole intervals while they fit
      while (!e.role_looped && s < p.K && used == 0 && cnt + p.nk[s] <= T) {
        cnt += p.nk[s];
        ++s;
      }
      if (cnt == 0) {  // interval larger than a tile (or partly consumed): take a slice of it
        const int left = p.nk[s] - used;
        cnt = left < T ? left : T;
        used += cnt;
        if (used == p.nk[s]) {
          used = 0;
          ++s;
        }
      }
      TileDev t;
      t.phase = ip;
      t.k0 = k;
      t.cnt = cnt;
      const NodeDev& a = p.nodes[k];
      const NodeDev& b = p.nodes[k + cnt - 1];
      t.span0 = a.dcol0;
      t.span_len = b.dcol0 + b.dlen - a.dcol0;
      t.drow0 = a.drow_off;  // phase-local, rebased below
      t.drow_len = b.drow_off + b.dlen - a.drow_off;
      const PhaseDev& q = e.phd[ip];
      t.N = q.N; t.phase_num = q.phase_num; t.x_state0 = q.x_state0; t.x_control0 = q.x_control0;
      t.x_t0 = q.x_t0; t.g0 = q.g0; t.v_nl0 = q.v_nl0; t.node0 = q.node0;
      t.c_src0 = t.c_cnt = t.c_dst0 = t.c_stride = t.c_copies = 0;  // filled once the phase's tile count is known
      e.tiles.push_back(t);
      if (t.span_len > e.max_span) e.max_span = t.span_len;
      if (t.drow_len > e.max_drow) e.max_drow = t.drow_len;
      k += cnt;
    }
    e.phd[ip].ntiles = int(e.tiles.size()) - e.phd[ip].tile0;
    // constant block: tile t of the phase copies sources [t*off/nt, (t+1)*off/nt) of the phase's Doffdiag list
    const PhaseDev& q = e.phd[ip];
    for (int t = 0; t < q.ntiles; ++t) {
      TileDev& tl = e.tiles[q.tile0 + t];
      // equal shares, cut at multiples of 16 entries so that every share starts on the same offset within a 128-byte line
      auto cut = [&](int tt) { return tt >= q.ntiles ? q.off_nnz : int(((long long)tt * q.off_nnz / q.ntiles) & ~15LL); };
      const int q0 = cut(t), q1 = cut(t + 1);
      tl.c_src0 = q.doff_base + q0;
      tl.c_cnt = q1 - q0;
      tl.c_dst0 = e.nnz_nl + e.nnz_lin + q.const_cum + q0;
      tl.c_stride = q.off_nnz;
      tl.c_copies = q.nx;
    }
  }
  // endpoint work items
  e.tasks.clear();
  e.tasks.push_back(TaskDev{0, 0});
  for (int ip = 0; ip < e.P; ++ip)
    if (e.ph[ip].ne > 0) e.tasks.push_back(TaskDev{1, ip});
  for (int i = 0; i < e.L; ++i) e.tasks.push_back(TaskDev{2, i});
  // rebase D-row offsets from phase-local to the concatenated dvals array
  {
    std::vector<int> base(e.P, 0);
    int acc = 0;
    for (int ip = 0; ip < e.P; ++ip) {
      base[ip] = acc;
      acc += int(e.ph[ip].drows.size());
    }
    for (TileDev& t : e.tiles) t.drow0 += base[t.phase];
  }
  // ownership: contiguous run of each phase's tiles per rank (interval sharding), balanced by node count
  e.my_tiles.clear();
  for (int ip = 0; ip < e.P; ++ip) {
    const int t0 = e.phd[ip].tile0, nt = e.phd[ip].ntiles;
    if (e.shard_mode != RPM_SHARD_INTERVALS || e.shard_world <= 1) {
      for (int t = 0; t < nt; ++t) e.my_tiles.push_back(t0 + t);
      continue;
    }
    const int N = e.ph[ip].N;
    for (int t = 0; t < nt; ++t) {
      const TileDev& tl = e.tiles[t0 + t];
      // owner of a tile = rank whose node range contains the tile's first node
      const long long lo = (long long)tl.k0 * e.shard_world / N;
      if (int(lo) == e.shard_rank) e.my_tiles.push_back(t0 + t);
    }
  }
}

int setup_engine(Engine& e, const rpm_problem_desc* d) {
  if (!d) return bad(e, RPM_E_INVALID, "null problem description");
  if (d->abi_version != RPM_ABI_VERSION) return bad(e, RPM_E_INVALID, "rpm_problem_desc.abi_version mismatch");
  if (d->n_phases < 1 || !d->phases) return bad(e, RPM_E_INVALID, "at least one phase is required");
  if (d->n_links < 0 || (d->n_links > 0 && !d->links)) return bad(e, RPM_E_INVALID, "bad linkage list");
  ProblemDims pd;
  if (!problem_dims(d->problem_id, &pd)) return bad(e, RPM_E_INVALID, "unknown problem_id");
  if (d->n_consts < pd.nconst) return bad(e, RPM_E_INVALID, "too few problem constants for this functor");
  if (d->first_derive == RPM_DERIVE_ANALYTIC && !pd.has_analytic)
    return bad(e, RPM_E_UNSUPPORTED, "first-derive=analytic: this problem functor has no analytic derivatives");
  if (d->n_instances < 1) return bad(e, RPM_E_INVALID, "n_instances must be >= 1");
  if (d->shard_world < 1 || d->shard_rank < 0 || d->shard_rank >= d->shard_world)
    return bad(e, RPM_E_INVALID, "bad shard rank/world");
  e.problem_id = d->problem_id;
  e.P = d->n_phases;
  e.L = d->n_links;
  e.consts.assign(d->consts, d->consts + d->n_consts);
  e.fd_tol = d->fd_tol > 0 ? d->fd_tol : 1e-6;
  e.first_derive = d->first_derive;
  e.hessian_mode = d->hessian_approximation;
  e.n_instances = d->n_instances;
  e.shard_mode = d->shard_mode;
  e.shard_rank = d->shard_rank;
  e.shard_world = d->shard_world;
  e.ph.assign(e.P, PhaseHost{});
  e.phd.assign(e.P, PhaseDev{});

  // ---- sizes + mesh checks -------------------------------------------------------------
  for (int i = 0; i < e.P; ++i) {
    const rpm_phase_desc& q = d->phases[i];
    PhaseHost& p = e.ph[i];
    const std::string tag = " in phase " + std::to_string(i + 1);
    p.nx = q.nx; p.nu = q.nu; p.nq = q.nq; p.nc = q.nc; p.ne = q.ne; p.K = q.n_intervals;
    if (q.nx != pd.nx || q.nu != pd.nu || q.nc != pd.nc)
      return bad(e, RPM_E_INVALID, "phase dimensions (nx,nu,nc) do not match the problem functor" + tag);
    if (q.ne < 0 || q.ne > pd.ne_max) return bad(e, RPM_E_INVALID, "too many event constraints for the functor" + tag);
    // static parameters: layout and derivative columns as the reference lays them out (LpBoundsChecker.cpp:117-138,
    // LpFiniteDifferenceDerive.cpp:299-317), values by the mathematically correct formulas where the reference's own
    // parameter path is inconsistent (DESIGN.md section 3 lists each departure)
    if (q.nq != pd.nq) return bad(e, RPM_E_INVALID, "the number of static parameters (nq) does not match the problem functor" + tag);
    if (q.nq > 0 && e.hessian_mode == RPM_HESSIAN_EXACT)
      return bad(e, RPM_E_UNSUPPORTED, "hessian-approximation=exact is not built for problems with static parameters (nq > 0): the reference's "
                                       "parameter Hessian is inconsistent (SURVEY.md App. A.6 quirks); use limited-memory" + tag);
    if (q.n_intervals < 1 || !q.mesh_points || !q.nodes_per_interval)
      return bad(e, RPM_E_INVALID, "MeshRefinement need at least two  meshPoints" + tag);
    if (q.mesh_points[0] != -1 || q.mesh_points[q.n_intervals] != 1)
      return bad(e, RPM_E_INVALID, "meshPoints must span -1 to +1" + tag);
    p.mesh.assign(q.mesh_points, q.mesh_points + q.n_intervals + 1);
    p.nk.assign(q.nodes_per_interval, q.nodes_per_interval + q.n_intervals);
    for (int s = 0; s < p.K; ++s) {
      if (p.nk[s] < 2) return bad(e, RPM_E_INVALID, "nodesPerInterval must be >= 2" + tag);
      if (!(p.mesh[s + 1] > p.mesh[s])) return bad(e, RPM_E_INVALID, "meshPoints must increase strictly" + tag);
    }
    build_tables(p);
    if (int(p.diag_v.size()) != p.N)
      return bad(e, RPM_E_INVALID, "differentiation matrix has a zero diagonal entry" + tag);
  }
  e.links.assign(e.L, LinkDev{});
  e.link_min.assign(e.L, {});
  e.link_max.assign(e.L, {});
  for (int i = 0; i < e.L; ++i) {
    const rpm_link_desc& q = d->links[i];
    if (q.left_phase < 1 || q.left_phase > e.P || q.right_phase < 1 || q.right_phase > e.P)
      return bad(e, RPM_E_INVALID, "The phase index is out of rang in linkage " + std::to_string(i + 1));
    if (q.n_links < 0 || q.n_links > pd.nlink_max)
      return bad(e, RPM_E_INVALID, "too many linkage constraints for the functor in linkage " + std::to_string(i + 1));
    e.links[i].left = q.left_phase - 1;
    e.links[i].right = q.right_phase - 1;
    e.links[i].nlink = q.n_links;
    e.link_min[i].assign(q.link_min, q.link_min + q.n_links);
    e.link_max[i].assign(q.link_max, q.link_max + q.n_links);
  }

  // ---- layout + bounds -------------------------------------------------------------------
  int n = 0, mnl = 0;
  for (int i = 0; i < e.P; ++i) {
    PhaseHost& p = e.ph[i];
    p.nvar = p.nx * (p.N + 1) + p.nu * p.N + 2 + p.nq;
    p.ncon = p.nx * p.N + p.nc * p.N + p.ne;
    p.var0 = n;
    p.con0 = mnl;
    n += p.nvar;
    mnl += p.ncon;
  }
  for (int i = 0; i < e.L; ++i) {
    e.links[i].g0 = mnl;
    mnl += e.links[i].nlink;
  }
  e.n = n;
  e.m_nl = mnl;
  e.m = mnl + e.P + e.L;
  e.xl.assign(n, 0.0); e.xu.assign(n, 0.0);
  e.gl.assign(e.m, 0.0); e.gu.assign(e.m, 0.0);
  int vi = 0, ci = 0;
  for (int i = 0; i < e.P; ++i) {
    const rpm_phase_desc& q = d->phases[i];
    const PhaseHost& p = e.ph[i];
    const std::string tag = std::to_string(i + 1);
    for (int j = 0; j < p.nx; ++j) {
      const double* mn = q.state_min + 3 * j;
      const double* mx = q.state_max + 3 * j;
      if (!(mn[0] <= mx[0] && mn[1] <= mx[1] && mn[2] <= mx[2]))
        return bad(e, RPM_E_INVALID, "Bounds on State are Inconsistent (i.e. max < min) in Phase:" + tag);
      e.xl[vi] = mn[0]; e.xu[vi++] = mx[0];
      e.gl[ci] = 0; e.gu[ci++] = 0;
      for (int k = 1; k < p.N; ++k) {
        e.xl[vi] = mn[1]; e.xu[vi++] = mx[1];
        e.gl[ci] = 0; e.gu[ci++] = 0;
      }
      e.xl[vi] = mn[2]; e.xu[vi++] = mx[2];
    }
    for (int j = 0; j < p.nu; ++j) {
      if (!(q.control_min[j] <= q.control_max[j]))
        return bad(e, RPM_E_INVALID, "Bounds on Control are Inconsistent (i.e. max < min) in Phase:" + tag);
      for (int k = 0; k < p.N; ++k) {
        e.xl[vi] = q.control_min[j]; e.xu[vi++] = q.control_max[j];
      }
    }
    e.xl[vi] = q.t0_min; e.xu[vi++] = q.t0_max;
    e.xl[vi] = q.tf_min; e.xu[vi++] = q.tf_max;
    for (int j = 0; j < p.nq; ++j) {   // LpBoundsChecker.cpp:117-138
      if (!q.parameter_min || !q.parameter_max || !(q.parameter_min[j] <= q.parameter_max[j]))
        return bad(e, RPM_E_INVALID, "Bounds on parameter are Inconsistent (i.e. max < min) in Phase:" + tag);
      e.xl[vi] = q.parameter_min[j]; e.xu[vi++] = q.parameter_max[j];
    }
    for (int j = 0; j < p.nc; ++j) {
      if (!(q.path_min[j] <= q.path_max[j]))
        return bad(e, RPM_E_INVALID, "Bounds on path are Inconsistent (i.e. max < min) in Phase:" + tag);
      for (int k = 0; k < p.N; ++k) {
        e.gl[ci] = q.path_min[j]; e.gu[ci++] = q.path_max[j];
      }
    }
    for (int j = 0; j < p.ne; ++j) {
      if (!(q.event_min[j] <= q.event_max[j]))
        return bad(e, RPM_E_INVALID, "Bounds on event are Inconsistent (i.e. max < min) in Phase:" + tag);
      e.gl[ci] = q.event_min[j]; e.gu[ci++] = q.event_max[j];
    }
  }
  for (int i = 0; i < e.L; ++i)
    for (int j = 0; j < e.links[i].nlink; ++j) {
      if (!(e.link_min[i][j] <= e.link_max[i][j]))
        return bad(e, RPM_E_INVALID, "Bounds on linkage are Inconsistent (i.e. max < min) in linkage:" + std::to_string(i + 1));
      e.gl[ci] = e.link_min[i][j]; e.gu[ci++] = e.link_max[i][j];
    }
  // linear rows: phase durations, then continuity of time across linked phases
  e.alin_i.clear(); e.alin_j.clear(); e.alin_v.clear();
  for (int i = 0; i < e.P; ++i) {
    const rpm_phase_desc& q = d->phases[i];
    const PhaseHost& p = e.ph[i];
    const int t0 = p.var0 + p.nx * (p.N + 1) + p.nu * p.N;
    e.alin_i.push_back(i); e.alin_j.push_back(t0); e.alin_v.push_back(-1);
    e.alin_i.push_back(i); e.alin_j.push_back(t0 + 1); e.alin_v.push_back(1);
    if (q.has_duration) {
      if (!(q.duration_min <= q.duration_max))
        return bad(e, RPM_E_INVALID, "Bounds on duration are Inconsistent (i.e. max < min) in Phase:" + std::to_string(i + 1));
      e.gl[e.m_nl + i] = q.duration_min;
      e.gu[e.m_nl + i] = q.duration_max;
    } else {
      e.gl[e.m_nl + i] = 0;
      e.gu[e.m_nl + i] = std::numeric_limits<double>::infinity();
    }
  }
  for (int i = 0; i < e.L; ++i) {
    const PhaseHost& pl = e.ph[e.links[i].left];
    const PhaseHost& pr = e.ph[e.links[i].right];
    e.alin_i.push_back(e.P + i); e.alin_j.push_back(pl.var0 + pl.nx * (pl.N + 1) + pl.nu * pl.N + 1); e.alin_v.push_back(-1);
    e.alin_i.push_back(e.P + i); e.alin_j.push_back(pr.var0 + pr.nx * (pr.N + 1) + pr.nu * pr.N); e.alin_v.push_back(1);
    e.gl[e.m_nl + e.P + i] = 0;
    e.gu[e.m_nl + e.P + i] = 0;
  }

  // ---- starting point ------------------------------------------------------------------
  e.guess.assign(n, 0.0);
  for (int i = 0; i < e.P; ++i) {
    const rpm_phase_desc& q = d->phases[i];
    const PhaseHost& p = e.ph[i];
    const int ng = q.n_guess;
    const std::string tag = std::to_string(i + 1);
    if (ng < 2 || !q.time_guess) return bad(e, RPM_E_INVALID, "Guess  must have a least two points in Phase:" + tag);
    for (int k = 1; k < ng; ++k)
      if (q.time_guess[k] == q.time_guess[0])
        return bad(e, RPM_E_INVALID, "Guess for time  does not contain unique valuesin phase " + tag);
    if ((p.nx > 0 && !q.state_guess) || (p.nu > 0 && !q.control_guess))
      return bad(e, RPM_E_INVALID, "Number of states in guess does not match limits in phase " + tag);
    const double t0g = q.time_guess[0], tfg = q.time_guess[ng - 1];
    std::vector<double> tau(ng);
    for (int k = 0; k < ng; ++k) tau[k] = 2 * (q.time_guess[k] - t0g) / (tfg - t0g) - 1;
    double* g = &e.guess[p.var0];
    int r = 0;
    for (int j = 0; j < p.nx; ++j) {
      for (int k = 0; k < p.N; ++k) g[r++] = spline_eval(p.points[k], tau.data(), q.state_guess + size_t(j) * ng, ng);
      g[r++] = spline_eval(1.0, tau.data(), q.state_guess + size_t(j) * ng, ng);
    }
    for (int j = 0; j < p.nu; ++j)
      for (int k = 0; k < p.N; ++k) g[r++] = spline_eval(p.points[k], tau.data(), q.control_guess + size_t(j) * ng, ng);
    g[r++] = t0g;
    g[r++] = tfg;
    if (p.nq > 0 && !q.parameter_guess)
      return bad(e, RPM_E_INVALID, "Number of parameters in guess does not match limits in phase " + tag);
    for (int j = 0; j < p.nq; ++j) g[r++] = q.parameter_guess[j];   // LpGuessChecker.cpp:186-189
  }

  // ---- Jacobian layout: values = [NL | LIN | CONST]  (LpNLPWrapper.cpp:244-252) -----------
  int node0 = 0, doff0 = 0, v = 0, cc = 0;
  for (int i = 0; i < e.P; ++i) {
    PhaseHost& p = e.ph[i];
    PhaseDev& q = e.phd[i];
    q.N = p.N; q.nx = p.nx; q.nu = p.nu; q.nc = p.nc; q.ne = p.ne; q.nq = p.nq;
    q.phase_num = i + 1;
    q.x_state0 = p.var0;
    q.x_control0 = p.var0 + p.nx * (p.N + 1);
    q.x_t0 = q.x_control0 + p.nu * p.N;
    q.g0 = p.con0;
    q.v_nl0 = v;
    v += (p.nx + p.nc) * (p.nx + p.nu + 2 + p.nq) * p.N;   // dependencies.fill(1): every block present (:1345); :682-683
    q.v_evt0 = v;
    v += p.ne * (2 * p.nx + 2 + p.nq);
    q.node0 = node0;
    node0 += p.N;
    q.doff_base = doff0;
    q.off_nnz = int(p.off_v.size());
    doff0 += q.off_nnz;
    q.const_cum = cc;
    cc += q.off_nnz * p.nx;
  }
  for (int i = 0; i < e.L; ++i) {
    LinkDev& l = e.links[i];
    if (e.ph[l.left].nx != e.ph[l.right].nx)
      return bad(e, RPM_E_UNSUPPORTED, "linked phases must have the same number of states (SURVEY.md B-11)");
    if (e.ph[l.left].nq != e.ph[l.right].nq)
      return bad(e, RPM_E_UNSUPPORTED, "linked phases must have the same number of static parameters (SURVEY.md B-11)");
    l.v0 = v;
    // columns [xf_left, p_left, x0_right, p_right] (:461-519); the reference counts the LEFT phase's sizes twice (B-11),
    // which is the same number whenever the linked phases have equal nx and nq
    v += l.nlink * (e.ph[l.left].nx + e.ph[l.left].nq + e.ph[l.right].nx + e.ph[l.right].nq);
  }
  e.nnz_nl = v;
  e.nnz_lin = int(e.alin_v.size());
  e.nnz_const = cc;
  e.nnz_jac = e.nnz_nl + e.nnz_lin + e.nnz_const;

  // ---- Jacobian structure (cached like GetConsSparsity's statics, but per engine) ----------
  e.jac_i.assign(e.nnz_jac, 0);
  e.jac_j.assign(e.nnz_jac, 0);
  {
    int s = 0, sc = e.nnz_nl + e.nnz_lin;
    for (int ip = 0; ip < e.P; ++ip) {
      const PhaseHost& p = e.ph[ip];
      const int N = p.N, nx = p.nx, nu = p.nu, nc = p.nc, ne = p.ne, nq = p.nq, disc = N + 1;
      const int r0 = p.con0, c0 = p.var0;
      auto diag_block = [&](int rs, int cs) {
        for (int k = 0; k < N; ++k) { e.jac_i[s] = r0 + rs + k; e.jac_j[s++] = c0 + cs + k; }
      };
      auto col_block = [&](int rs, int col) {
        for (int k = 0; k < N; ++k) { e.jac_i[s] = r0 + rs + k; e.jac_j[s++] = c0 + col; }
      };
      for (int o = 0; o < nx + nc; ++o) {
        const int rs = o * N;   // defect rows i*N, then path rows nx*N + i*N
        for (int j = 0; j < nx; ++j) diag_block(rs, j * disc);
        for (int j = 0; j < nu; ++j) diag_block(rs, nx * disc + j * N);
        col_block(rs, nx * disc + nu * N);
        col_block(rs, nx * disc + nu * N + 1);
        // one block per static parameter: every node's row, the parameter's ONE column (the reference writes a diagonal
        // run `indexvector + colstart` here, LpNLPWrapper.cpp:1211,1262 — SURVEY B-8; not reproduced)
        for (int j = 0; j < nq; ++j) col_block(rs, nx * disc + nu * N + 2 + j);
      }
      for (int i = 0; i < ne; ++i) {
        const int row = r0 + (nx + nc) * N + i;
        for (int j = 0; j < nx; ++j) {
          e.jac_i[s] = row; e.jac_j[s++] = c0 + N * j + j;
          e.jac_i[s] = row; e.jac_j[s++] = c0 + N * (j + 1) + j;
        }
        e.jac_i[s] = row; e.jac_j[s++] = c0 + nx * disc + nu * N;
        e.jac_i[s] = row; e.jac_j[s++] = c0 + nx * disc + nu * N + 1;
        for (int j = 0; j < nq; ++j) { e.jac_i[s] = row; e.jac_j[s++] = c0 + nx * disc + nu * N + 2 + j; }   // :854-859
      }
      for (int i = 0; i < nx; ++i)
        for (size_t q = 0; q < p.off_v.size(); ++q) {
          e.jac_i[sc] = r0 + i * N + p.off_i[q];
          e.jac_j[sc++] = c0 + i * disc + p.off_j[q];
        }
    }
    for (int ip = 0; ip < e.L; ++ip) {
      const LinkDev& l = e.links[ip];
      const PhaseHost& pl = e.ph[l.left];
      const PhaseHost& pr = e.ph[l.right];
      for (int jc = 0; jc < pl.nx; ++jc)
        for (int ir = 0; ir < l.nlink; ++ir) { e.jac_i[s] = l.g0 + ir; e.jac_j[s++] = pl.var0 + (jc + 1) * pl.N + jc; }
      for (int jc = 0; jc < pl.nq; ++jc)
        for (int ir = 0; ir < l.nlink; ++ir) { e.jac_i[s] = l.g0 + ir; e.jac_j[s++] = pl.var0 + pl.nx * (pl.N + 1) + pl.nu * pl.N + 2 + jc; }
      for (int jc = 0; jc < pr.nx; ++jc)
        for (int ir = 0; ir < l.nlink; ++ir) { e.jac_i[s] = l.g0 + ir; e.jac_j[s++] = pr.var0 + jc * (pr.N + 1); }
      for (int jc = 0; jc < pr.nq; ++jc)
        for (int ir = 0; ir < l.nlink; ++ir) { e.jac_i[s] = l.g0 + ir; e.jac_j[s++] = pr.var0 + pr.nx * (pr.N + 1) + pr.nu * pr.N + 2 + jc; }
    }
    for (int q = 0; q < e.nnz_lin; ++q) {
      e.jac_i[e.nnz_nl + q] = e.m_nl + e.alin_i[q];
      e.jac_j[e.nnz_nl + q] = e.alin_j[q];
    }
  }

  // ---- concatenated device tables ----------------------------------------------------------
  e.nodes.clear(); e.points.clear(); e.weights.clear(); e.diag.clear(); e.dvals.clear(); e.doff_vals.clear();
  for (int i = 0; i < e.P; ++i) {
    const PhaseHost& p = e.ph[i];
    const int base = int(e.dvals.size());
    for (NodeDev nd : p.nodes) {
      nd.drow_off += base;
      e.nodes.push_back(nd);
    }
    e.points.insert(e.points.end(), p.points.begin(), p.points.end());
    e.weights.insert(e.weights.end(), p.weights.begin(), p.weights.end());
    e.diag.insert(e.diag.end(), p.diag_v.begin(), p.diag_v.end());
    e.dvals.insert(e.dvals.end(), p.drows.begin(), p.drows.end());
    e.doff_vals.insert(e.doff_vals.end(), p.off_v.begin(), p.off_v.end());
  }
  // tiling.  Small grids (one iterate per launch): 16 nodes x (nx+nu+2) roles per workgroup, every role its own
  // thread — shortest critical path.  Large grids (>= 2 workgroups per CU even with 64-node tiles): the role-looped
  // layout, 64 nodes x 4 role groups (rpm_tile_rl_kernel) — 512-byte store runs, one residency round.
  int T = e.opt_tile_nodes;
  e.role_looped = false;
  if (T != 16 && T != 32 && T != 64) {
    long long total = 0;
    for (int i = 0; i < e.P; ++i) total += e.ph[i].N;
    total *= e.n_instances;
    if (e.opt_role_loop != 0 && (e.opt_role_loop == 1 || total / 64 >= 512)) {
      T = 64;
      e.role_looped = true;
    } else {
      T = 16;
    }
  }
  build_tiles(e, T);
  return RPM_OK;
}

}  // namespace rpm

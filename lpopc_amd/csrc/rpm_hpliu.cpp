// rpm_hpliu.cpp — hp-Liu mesh refinement on the host (second method): everything here is
// history-dependent integer / small-matrix logic per mesh interval; the only per-node work it needs, the relative-error
// matrix, comes from the device (dev_solution_error).  Reference: LiuHpMeshRefineAlg,
// /root/reference/Lpopc/src/Core/LpLiuHpMeshRefineAlg.cpp —
//   RefineMesh :12-260 · Lagrange power-series coefficients :263-304 · Dividing_mesh :321-377 · Increasing_N :379-436 ·
//   Reducing_N :438-481 · Merging_mesh :483-604 (verdict unused by RefineMesh: equal-N satisfied neighbours always
//   merge, :197-220) · CanWeIncreaseN :606-681 · calculate2nd_derive :683-709.
// Kept as written, including the mismatched abscissae of the second-derivative sampling (:689-699) and the pairing of
// the current mesh points with rows of the previous solution in CanWeIncreaseN (:649-660).  Where the reference would
// throw (empty find(), row past the previous state matrix) or cast a non-finite double to uword, refine() fails with
// RPM_E_INVALID and a message.  One deviation: a phase that is fully satisfied while an earlier phase still refines keeps
// its mesh in the history (the reference stores a null mesh there and dereferences it on the next call, :159).
#include <cmath>
#include <cstring>
#include <limits>

#include "rpm_engine.hpp"

namespace rpm {
namespace {

enum Tag { kNotSatisfied, kSatisfied, kReduced, kMerged };

// two interleaved running sums (arma::accu / sum of a vector)
double pair_sum(const std::vector<double>& a) {
  double s0 = 0.0, s1 = 0.0;
  size_t i = 0;
  for (; i + 1 < a.size(); i += 2) {
    s0 += a[i];
    s1 += a[i + 1];
  }
  if (i < a.size()) s0 += a[i];
  return s0 + s1;
}

// barycentric interpolation of (xs, ys) at xq (SolutionErrorChecker::BarLagrangeInterp)
void interpolate(const std::vector<double>& xs, const std::vector<double>& ys, const std::vector<double>& xq,
                 std::vector<double>& out) {
  const int m = int(xs.size()), nq = int(xq.size());
  std::vector<double> H(size_t(m) * nq), S(nq);
  std::vector<int> hit(nq);
  lagrange_rows(xs.data(), m, xq.data(), nq, H.data(), S.data(), hit.data());
  out.resize(nq);
  for (int r = 0; r < nq; ++r) {
    double acc = 0.0;
    for (int j = 0; j < m; ++j) acc += H[r + size_t(j) * nq] * ys[j];
    out[r] = hit[r] >= 0 ? ys[hit[r]] : acc / S[r];
  }
}

// descending power-series coefficients of the Lagrange basis on [LGR(N); 1]; column i belongs to node i
std::vector<double> basis_coefficients(int N) {
  const int M = N + 1;
  std::vector<double> x, w;
  lgr_points(N, x, w);
  x.push_back(1.0);
  std::vector<double> out(size_t(M) * M), sym(size_t(N) * N), roots(N), D(M), pw(M), pd(M);
  for (int i = 0; i < M; ++i) {
    int q = 0;
    for (int j = 0; j < M; ++j)
      if (j != i) roots[q++] = -x[j];
    std::fill(sym.begin(), sym.end(), 0.0);   // elementary symmetric sums, row r = degree r + 1
    for (int j = 0; j < N; ++j) sym[size_t(j) * N] = roots[j];
    for (int r = 1; r < N; ++r) {
      for (int j = N - 2; j >= 0; --j) sym[r + size_t(j) * N] = sym[r + size_t(j + 1) * N] + sym[(r - 1) + size_t(j + 1) * N];
      for (int j = 0; j < N; ++j) sym[r + size_t(j) * N] *= roots[j];
    }
    D[0] = 1.0;
    for (int r = 0; r < N; ++r) {
      double s = sym[r];
      for (int j = 1; j < N; ++j) s += sym[r + size_t(j) * N];
      D[r + 1] = s;
    }
    pw[N] = 1.0;
    for (int k = N - 1; k >= 0; --k) pw[k] = pw[k + 1] * x[i];
    for (int k = 0; k < M; ++k) pd[k] = pw[k] * D[k];
    const double den = pair_sum(pd);
    for (int k = 0; k < M; ++k) out[k + size_t(i) * M] = D[k] / den;
  }
  return out;
}

int nodes_needed(double tol, int N, const double* seg, int ld, int nx, const std::vector<double>& beta) {
  const int M = N + 1;
  const std::vector<double> a = basis_coefficients(N);
  int best = 0;
  for (int s = 0; s < nx; ++s) {
    int first = -1;
    for (int r = 0; r < M && first < 0; ++r) {
      double b = 0.0;
      for (int k = 0; k < M; ++k) b += a[r + size_t(k) * M] * seg[k + size_t(s) * ld];
      if (b / beta[s] > tol) first = r;
    }
    best = std::max(best, first < 0 ? 1 : M - 1 - first);
  }
  return std::max(best, 2);
}

// largest |second difference| of every state on 501 samples and where it sits
void curvature(const double* t, int n, const double* x, int ld, int nx, std::vector<double>& peak, std::vector<double>& at) {
  const double t0 = t[0], tf = t[n - 1];
  std::vector<double> tau(n), tp(501), col(n), xp;
  for (int i = 0; i < n; ++i) tau[i] = 2.0 * (t[i] - t0) / (tf - t0) - 1.0;
  const double step = 2.0 / 500.0, delta = (tf - t0) / 500.0;
  for (int i = 0; i < 500; ++i) tp[i] = t0 + i * delta;
  tp[500] = tf;
  peak.assign(nx, 0.0);
  at.assign(nx, 0.0);
  for (int s = 0; s < nx; ++s) {
    for (int i = 0; i < n; ++i) col[i] = x[i + size_t(s) * ld];
    interpolate(tau, col, tp, xp);
    double best = -1.0;
    int bi = 0;
    for (int i = 0; i < 499; ++i) {
      double d = (xp[i + 2] - 2 * xp[i + 1]) + xp[i];
      d /= (step * step);
      d = std::fabs(d);
      if (d > best) {
        best = d;
        bi = i;
      }
    }
    peak[s] = best < 0 ? -std::numeric_limits<double>::infinity() : best;
    at[s] = tp[bi];
  }
}

int last_below(const std::vector<double>& a, double v, bool strict) {
  int r = -1;
  for (int i = 0; i < int(a.size()); ++i)
    if (strict ? a[i] < v : a[i] <= v) r = i;
  return r;
}
int first_above(const std::vector<double>& a, double v, bool strict) {
  for (int i = 0; i < int(a.size()); ++i)
    if (strict ? a[i] > v : a[i] >= v) return i;
  return -1;
}
long as_count(double v) { return (v != v || v < 0 || v > 1e9) ? -1 : long(v); }

}  // namespace

int HpLiu::smooth_enough(int ip, int first_row, int n, const std::vector<double>& tau, const double* state, int ld, int nx) const {
  std::vector<double> pk, at, pk_b, at_b;
  curvature(tau.data() + first_row, n + 1, state + first_row, ld, nx, pk, at);
  double lo_t = at[0], hi_t = at[0];
  for (int s = 1; s < nx; ++s) {
    lo_t = std::min(lo_t, at[s]);
    hi_t = std::max(hi_t, at[s]);
  }
  const std::vector<double>& mp = meshes.back()[ip].mesh;
  int lo, hi;
  if (lo_t == hi_t) {
    if (lo_t == tau[first_row]) { lo = last_below(mp, lo_t, false); hi = first_above(mp, hi_t, true); }
    else if (hi_t == tau[first_row + n]) { lo = last_below(mp, lo_t, true); hi = first_above(mp, hi_t, false); }
    else { lo = last_below(mp, lo_t, true); hi = first_above(mp, hi_t, true); }
  } else {
    lo = last_below(mp, lo_t, false);
    hi = first_above(mp, hi_t, false);
  }
  const std::vector<double>& tb = points_hist.back()[ip];
  const Solution& sb = states.back()[ip];
  if (lo < 0 || hi < 0 || hi < lo || hi >= int(tb.size()) || hi >= sb.rows) return -1;
  curvature(tb.data() + lo, hi - lo + 1, sb.v.data() + lo, sb.rows, nx, pk_b, at_b);
  double worst = -std::numeric_limits<double>::infinity();
  for (int s = 0; s < nx; ++s) {
    const double r = pk[s] / pk_b[s];
    if (r > worst) worst = r;
  }
  return worst > R ? 0 : 1;
}

bool HpLiu::exponent(int ip, double m0, double mf, int N, double e_k, double* q) const {
  const Mesh& b = meshes[meshes.size() - 2][ip];
  const int lo = last_below(b.mesh, m0, false), hi = first_above(b.mesh, mf, false);
  if (lo < 0 || hi < 0 || hi - 1 < lo) return false;
  const double h = mf - m0, hb = b.mesh[hi] - b.mesh[lo];
  int Nb = 0;
  double eb = b.e_k[lo];
  for (int i = lo; i < hi; ++i) {
    Nb += b.nodes[i];
    eb = std::max(eb, b.e_k[i]);
  }
  const double fN = double(N) / double(Nb), fh = h / hb, fe = e_k / eb;
  *q = std::ceil(std::log(fe / std::pow(double(N), 5.0 / 2.0)) / std::log(fh / fN));
  return true;
}

// One LiuHpMeshRefineAlg::RefineMesh over all phases.  rel[p]: relative-error matrix of phase p; x: the NLP solution.
int HpLiu::refine(const Engine& e, const double* x, const std::vector<std::vector<double>>& rel,
                  std::vector<std::vector<double>>& new_mesh, std::vector<std::vector<int>>& new_nodes, bool* no_more,
                  std::string* why) {
  const int P = e.P;
  if (mesh_index == 0) {
    std::vector<Mesh> first(P);
    for (int p = 0; p < P; ++p) {
      first[p].mesh = e.ph[p].mesh;
      first[p].nodes = e.ph[p].nk;
      first[p].e_k.assign(e.ph[p].K, 0.0);
    }
    meshes.push_back(first);
  }
  std::vector<Mesh>& before = meshes.back();
  std::vector<Mesh> out(P);
  std::vector<Solution> sol(P);
  std::vector<std::vector<double>> pts(P);
  bool done = true;
  new_mesh.assign(P, {});
  new_nodes.assign(P, {});
  auto fail = [&](const char* msg) {
    if (mesh_index == 0) meshes.pop_back();
    *why = msg;
    return RPM_E_INVALID;
  };
  for (int ip = 0; ip < P; ++ip) {
    const PhaseHost& p = e.ph[ip];
    const int K = p.K, N = p.N, nx = p.nx, M1 = N + 1, rows = N + K + 1;
    if (int(before[ip].nodes.size()) != K) return fail("hp-Liu: the engine is not built on the mesh this object produced last");
    const double* state = x + e.phd[ip].x_state0;
    const double* err = rel[ip].data();
    std::vector<double> tau(p.points);
    tau.push_back(1.0);
    sol[ip].rows = M1;
    sol[ip].v.assign(state, state + size_t(M1) * nx);
    std::vector<double> beta(nx);
    for (int s = 0; s < nx; ++s) {
      double mx = state[size_t(s) * M1];
      for (int r = 1; r < M1; ++r) mx = std::max(mx, state[r + size_t(s) * M1]);
      beta[s] = 1 + mx;
    }
    struct Seg { double m0, mf; int parts, nodes, tag; };
    std::vector<Seg> segs(K);
    int row_e = 0, row_x = 0;
    for (int k = 0; k < K; ++k) {
      const int n = p.nk[k], last = row_e + n + 1;
      double emax = err[row_e];
      for (int s = 0; s < nx; ++s)
        for (int r = row_e; r <= last; ++r) emax = std::max(emax, err[r + size_t(s) * rows]);
      before[ip].e_k[k] = emax;
      Seg sg{p.mesh[k], p.mesh[k + 1], 1, n, kNotSatisfied};
      if (emax <= tol) {
        sg.nodes = nodes_needed(tol, n, state + row_x, M1, nx, beta);
        sg.tag = sg.nodes == n ? kSatisfied : kReduced;
        if (sg.tag == kReduced) done = false;
      } else {
        if (mesh_index == 0) {
          sg.nodes = n + 3;
        } else {
          const int smooth = smooth_enough(ip, row_x, n, tau, state, M1, nx);
          if (smooth < 0) return fail("hp-Liu: CanWeIncreaseN indexes outside the previous mesh/solution (the reference throws here)");
          bool divide = !smooth;
          double q = 0.0;
          if (smooth) {
            if (!exponent(ip, sg.m0, sg.mf, n, emax, &q)) return fail("hp-Liu: interval not found in the previous mesh (the reference throws here)");
            const long need = as_count(std::ceil(n * std::pow(emax / tol, 1.0 / (q - 5.0 / 2.0))));
            if (need < 0 || need > Nmax) divide = true;
            else sg.nodes = int(need);
          }
          if (divide) {
            if (!exponent(ip, sg.m0, sg.mf, n, emax, &q)) return fail("hp-Liu: interval not found in the previous mesh (the reference throws here)");
            const long H = as_count(std::ceil(std::pow(emax / tol, 1 / q)));
            const long Hmax = as_count(std::ceil(std::log(emax / tol) / std::log(double(n))));
            if (H < 0 && Hmax < 0) return fail("hp-Liu: the number of sub-intervals is not finite (undefined cast in the reference)");
            long S = H < 0 ? Hmax : Hmax < 0 ? H : std::min(H, Hmax);
            sg.parts = int(std::max(S, 2L));
            sg.nodes = n;
          }
        }
        sg.tag = kNotSatisfied;
        done = false;
      }
      segs[k] = sg;
      row_e = last;
      row_x += n;
    }
    if (!done) {   // merge equal-N neighbours that need no refinement
      size_t idx = 0;
      for (int k = 0; k < K; ++k) {
        if (k > 0 && segs[idx].tag != kNotSatisfied && segs[idx - 1].tag != kNotSatisfied && segs[idx].nodes == segs[idx - 1].nodes) {
          segs[idx - 1].mf = segs[idx].mf;
          segs[idx - 1].tag = kMerged;
          segs.erase(segs.begin() + idx);
        } else {
          ++idx;
        }
      }
    }
    std::vector<double>& om = new_mesh[ip];
    std::vector<int>& on = new_nodes[ip];
    om.push_back(-1);
    for (const Seg& sg : segs) {
      const double delta = (sg.mf - sg.m0) / double(sg.parts);
      for (int i = 1; i <= sg.parts; ++i) {
        om.push_back(i == sg.parts ? sg.mf : sg.m0 + i * delta);
        on.push_back(sg.nodes);
      }
    }
    out[ip].mesh = om;
    out[ip].nodes = on;
    out[ip].e_k.assign(on.size(), 0.0);
    pts[ip] = om;
  }
  meshes.push_back(out);
  states.push_back(sol);
  points_hist.push_back(pts);
  ++mesh_index;
  *no_more = done;
  return RPM_OK;
}

}  // namespace rpm

// rpm_mesh.cpp — host side of the mesh-error estimate and the ph refinement decision.
// The tables depend only on the mesh, like D: they are built once per engine; the per-node work (interpolation,
// dynamics, integration, error) runs on the device (rpm_mesh_err_kernel in rpm_device.hip).
//   SolutionErrorChecker::BarLagrangeInterp / SolutionInterpolation   /root/reference/Lpopc/src/Core/LpSolutionError.cpp:10-108
//   RPMGenerator IntegrationMatrix = inv(D(:,1:)) per interval          Core/RPMGenerator.cpp:85-104,200-251
//   PhMeshRefineAlg::RefineMesh / ModifySegment                        Core/LpPhMeshRefineAlg.cpp:12-100
#include <algorithm>
#include <cmath>

#include "rpm_engine.hpp"

namespace rpm {
namespace {

// product of a column the way arma::prod does it (two interleaved running products)
double pair_product(const std::vector<double>& a) {
  double p0 = 1.0, p1 = 1.0;
  size_t i = 0;
  for (; i + 1 < a.size(); i += 2) {
    p0 *= a[i];
    p1 *= a[i + 1];
  }
  if (i < a.size()) p0 *= a[i];
  return p0 * p1;
}

}  // namespace

// barycentric-form interpolation rows from `src` (m points) to `dst` (nq points): H is nq x m column-major with
// H(r,j) = w_j / (dst_r - src_j); S its row sums; hit[r] = j when dst_r coincides with src_j (the value is copied)
void lagrange_rows(const double* src, int m, const double* dst, int nq, double* H, double* S, int* hit) {
  std::vector<double> w(m), col(m);
  for (int j = 0; j < m; ++j) {
    for (int i = 0; i < m; ++i) col[i] = (src[i] - src[j]) + (i == j ? 1.0 : 0.0);
    w[j] = 1 / pair_product(col);
  }
  std::fill(hit, hit + nq, -1);
  for (int r = 0; r < nq; ++r) {
    double sum = 0.0;
    for (int j = 0; j < m; ++j) {
      double dist = dst[r] - src[j];
      if (dist == 0) {
        hit[r] = j;
        dist = std::nan("");
      }
      const double h = w[j] / dist;
      H[r + size_t(j) * nq] = h;
      sum = (j == 0) ? h : sum + h;
    }
    S[r] = sum;
  }
}

namespace {

// dense inverse, Gaussian elimination with row pivoting, then one forward/back substitution per unit vector
void invert(int n, std::vector<double> a, double* out) {
  std::vector<int> swap_with(n);
  auto at = [&](int i, int j) -> double& { return a[i + size_t(j) * n]; };
  for (int k = 0; k < n; ++k) {
    int best = k;
    double big = std::fabs(at(k, k));
    for (int i = k + 1; i < n; ++i)
      if (std::fabs(at(i, k)) > big) {
        big = std::fabs(at(i, k));
        best = i;
      }
    swap_with[k] = best;
    if (best != k)
      for (int j = 0; j < n; ++j) std::swap(at(k, j), at(best, j));
    for (int i = k + 1; i < n; ++i) {
      at(i, k) /= at(k, k);
      const double l = at(i, k);
      for (int j = k + 1; j < n; ++j) at(i, j) -= l * at(k, j);
    }
  }
  for (int c = 0; c < n; ++c) {
    double* b = out + size_t(c) * n;
    for (int i = 0; i < n; ++i) b[i] = (i == c) ? 1.0 : 0.0;
    for (int k = 0; k < n; ++k)
      if (swap_with[k] != k) std::swap(b[k], b[swap_with[k]]);
    for (int i = 0; i < n; ++i)
      for (int j = 0; j < i; ++j) b[i] -= at(i, j) * b[j];
    for (int i = n - 1; i >= 0; --i) {
      for (int j = i + 1; j < n; ++j) b[i] -= at(i, j) * b[j];
      b[i] /= at(i, i);
    }
  }
}

}  // namespace

void build_mesh_err_tables(const PhaseHost& p, MeshErrTables& t) {
  t = MeshErrTables();
  const int N = p.N, K = p.K;
  std::vector<double> tau(p.points);
  tau.push_back(1.0);
  t.fine_nodes = N + K;
  t.rows = N + K + 1;
  int istart = 0, r0 = 0;
  for (int seg = 0; seg < K; ++seg) {
    const int n = p.nk[seg], n1 = n + 1;
    MeshIvDev iv;
    iv.n = n;
    iv.istart = istart;
    iv.r0 = r0;
    iv.q0 = int(t.ttem.size());
    iv.hs = int(t.Hs.size());
    iv.hc = int(t.Hc.size());
    iv.a = int(t.A.size());
    const double time0 = tau[istart], timef = tau[istart + n];
    std::vector<double> xi, wi;
    lgr_points(n1, xi, wi);
    std::vector<double> q(n1);
    for (int k = 0; k < n1; ++k) q[k] = (xi[k] + 1) * (timef - time0) / 2 + time0;   // LpSolutionError.cpp:75
    t.ttem.insert(t.ttem.end(), q.begin(), q.end());
    // states: n+1 data points (the interval's nodes and the next interval's first point) -> n+1 new points
    t.Hs.resize(t.Hs.size() + size_t(n1) * n1);
    t.Ss.resize(t.Ss.size() + n1);
    t.hit_s.resize(t.hit_s.size() + n1);
    lagrange_rows(tau.data() + istart, n1, q.data(), n1, t.Hs.data() + iv.hs, t.Ss.data() + iv.q0, t.hit_s.data() + iv.q0);
    // controls: the interval's n nodes -> the same n+1 new points
    t.Hc.resize(t.Hc.size() + size_t(n1) * n);
    t.Sc.resize(t.Sc.size() + n1);
    t.hit_c.resize(t.hit_c.size() + n1);
    lagrange_rows(tau.data() + istart, n, q.data(), n1, t.Hc.data() + iv.hc, t.Sc.data() + iv.q0, t.hit_c.data() + iv.q0);
    // integration matrix of the (n+1)-point interval, RPMGenerator.cpp:70-86 with nodesPerInterval + 1
    const double span = p.mesh[seg + 1] - p.mesh[seg];
    std::vector<double> pts(n1 + 1);
    for (int k = 0; k < n1; ++k) {
      double v = xi[k] + 1;
      v *= span / 2.0;
      v += p.mesh[seg];
      pts[k] = v;
    }
    pts[n1] = p.mesh[seg + 1];
    std::vector<double> D;
    colloc_d(pts, D);   // n1 x (n1+1)
    t.A.resize(t.A.size() + size_t(n1) * n1);
    invert(n1, std::vector<double>(D.begin() + n1, D.end()), t.A.data() + iv.a);
    t.iv.push_back(iv);
    istart += n;
    r0 += n1;
  }
}

// PhMeshRefineAlg::RefineMesh for one phase from its relative_error matrix (rows x nx, column-major)
bool ph_refine(const PhaseHost& p, const double* rel, double tol, int nmin, int nmax, std::vector<double>& mesh,
               std::vector<int>& nodes, std::vector<double>& interval_error) {
  const int rows = p.N + p.K + 1;
  bool no_more = true;
  mesh.assign(1, -1.0);
  nodes.clear();
  interval_error.assign(p.K, 0.0);
  int istart = 0;
  for (int seg = 0; seg < p.K; ++seg) {
    const int n = p.nk[seg], ifinish = istart + n + 1;
    double emax = rel[istart];
    for (int s = 0; s < p.nx; ++s)
      for (int r = istart; r <= ifinish; ++r) emax = std::max(emax, rel[r + size_t(s) * rows]);
    interval_error[seg] = emax;
    const double lo = p.mesh[seg], hi = p.mesh[seg + 1];
    if (emax <= tol) {
      mesh.push_back(hi);
      nodes.push_back(n);
    } else {
      no_more = false;
      const int grow = static_cast<int>(std::log(emax / tol) / std::log(double(n)));   // Pq, LpPhMeshRefineAlg.cpp:81
      const int want = n + grow;
      if (want <= nmax) {
        mesh.push_back(hi);
        nodes.push_back(want);
      } else {
        const int parts = static_cast<int>(std::max(std::ceil(double(want) / double(nmin)), 2.0));   // Bq, :94
        const double step = (hi - lo) / double(parts);   // arma::linspace: start + i*delta, last point = end
        for (int i = 1; i <= parts; ++i) {
          mesh.push_back(i == parts ? hi : lo + i * step);
          nodes.push_back(nmin);
        }
      }
    }
    istart = ifinish;
  }
  return no_more;
}

}  // namespace rpm

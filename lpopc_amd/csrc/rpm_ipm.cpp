// rpm_ipm.cpp — time-ordered band + border layout of the primal-dual KKT matrix (host side of row f-2, see rpm_ipm.hpp).
#include "rpm_ipm.hpp"

#include <algorithm>
#include <cstdint>
#include <cstdlib>

namespace rpm {

int build_ipm_plan(Engine& e, IpmPlan& p, std::string* why) {
  p = IpmPlan();
  p.n = e.n;
  p.m = e.m;
  if (int(e.hes_i.size()) != e.nnz_h || int(e.jac_i.size()) != e.nnz_jac) {
    if (why) *why = "the Jacobian / Hessian structure is not built yet";
    return RPM_E_INVALID;
  }
  // rows: equality when g_l == g_u, otherwise one slack each (bounds g_l <= s <= g_u)
  p.row_slack.assign(p.m, -1);
  for (int r = 0; r < p.m; ++r)
    if (e.gl[r] != e.gu[r]) {
      p.row_slack[r] = p.ns++;
      p.slack_row.push_back(r);
    }
  p.nv = p.n + p.ns;
  p.Nt = p.nv + p.m;
  p.fixed.assign(p.n, 0);
  for (int i = 0; i < p.n; ++i) p.fixed[i] = e.xl[i] == e.xu[i] ? 1 : 0;

  // time key of every unknown: global node index for what lives at a collocation node, -1 for the border
  std::vector<long long> key(p.Nt, -1);
  long long node_base = 0;
  for (int ip = 0; ip < e.P; ++ip) {
    const PhaseDev& q = e.phd[ip];
    for (int i = 0; i < q.nx; ++i)
      for (int k = 0; k < q.N; ++k) key[q.x_state0 + i * (q.N + 1) + k] = node_base + k;   // X(N, .) stays in the border
    for (int j = 0; j < q.nu; ++j)
      for (int k = 0; k < q.N; ++k) key[q.x_control0 + j * q.N + k] = node_base + k;
    for (int r = 0; r < (q.nx + q.nc) * q.N; ++r) {      // defects then paths, each N consecutive rows (LpNLPWrapper.cpp:138-228)
      const int row = q.g0 + r;
      key[p.nv + row] = node_base + r % q.N;
      if (p.row_slack[row] >= 0) key[p.n + p.row_slack[row]] = node_base + r % q.N;
    }
    node_base += q.N;
  }
  // order: by node, inside a node variables, slacks, multipliers; the border keeps variables, slacks, multipliers
  std::vector<int> band, border;
  for (int u = 0; u < p.Nt; ++u) (key[u] >= 0 ? band : border).push_back(u);
  auto type_of = [&](int u) { return u < p.n ? 0 : (u < p.nv ? 1 : 2); };
  std::stable_sort(band.begin(), band.end(), [&](int a, int c) {
    if (key[a] != key[c]) return key[a] < key[c];
    return type_of(a) < type_of(c);
  });
  p.Nb = int(band.size());
  p.nb = int(border.size());
  p.pos.assign(p.Nt, 0);
  for (int i = 0; i < p.Nb; ++i) p.pos[band[i]] = i;
  for (int i = 0; i < p.nb; ++i) p.pos[border[i]] = p.Nb + i;

  // half bandwidth over the entries that stay inside the band
  auto reach = [&](int ua, int uc) {
    const int a = p.pos[ua], c = p.pos[uc];
    if (a < p.Nb && c < p.Nb) p.b = std::max(p.b, std::abs(a - c));
  };
  for (int k = 0; k < e.nnz_jac; ++k)
    if (!p.fixed[e.jac_j[k]]) reach(p.nv + e.jac_i[k], e.jac_j[k]);
  for (int k = 0; k < e.nnz_h; ++k)
    if (!p.fixed[e.hes_i[k]] && !p.fixed[e.hes_j[k]]) reach(e.hes_i[k], e.hes_j[k]);
  for (int s = 0; s < p.ns; ++s) reach(p.nv + p.slack_row[s], p.n + s);
  p.b = std::max(p.b, 16);                       // a 16-column block of the factorisation fits inside the band
  if (p.b > p.Nb - 1) p.b = std::max(p.Nb - 1, 0);
  p.CS = p.b + 1 + p.nb;
  p.CS += p.CS & 1;

  auto dst = [&](int ua, int uc) -> int {
    int a = p.pos[ua], c = p.pos[uc];
    if (a < c) std::swap(a, c);
    const long long o = p.at(a, c);
    return o > INT32_MAX ? -2 : int(o);
  };
  bool too_big = p.storage() > INT32_MAX;
  p.jac_dst.assign(e.nnz_jac, -1);
  p.hes_dst.assign(e.nnz_h, -1);
  if (!too_big) {
    for (int k = 0; k < e.nnz_jac; ++k)
      if (!p.fixed[e.jac_j[k]]) p.jac_dst[k] = dst(p.nv + e.jac_i[k], e.jac_j[k]);
    for (int k = 0; k < e.nnz_h; ++k)
      if (!p.fixed[e.hes_i[k]] && !p.fixed[e.hes_j[k]]) p.hes_dst[k] = dst(e.hes_i[k], e.hes_j[k]);
    p.diag_dst.resize(p.Nt);
    for (int u = 0; u < p.Nt; ++u) p.diag_dst[u] = dst(u, u);
    p.slk_dst.resize(p.ns);
    for (int s = 0; s < p.ns; ++s) p.slk_dst[s] = dst(p.nv + p.slack_row[s], p.n + s);
  }
  if (too_big) {
    if (why) *why = "KKT storage of one instance exceeds 2^31 doubles";
    return RPM_E_UNSUPPORTED;
  }
  // Jacobian by column, entries of a column in COO order
  p.jt_ptr.assign(p.n + 1, 0);
  for (int k = 0; k < e.nnz_jac; ++k) ++p.jt_ptr[e.jac_j[k] + 1];
  for (int i = 0; i < p.n; ++i) p.jt_ptr[i + 1] += p.jt_ptr[i];
  p.jt_ent.resize(e.nnz_jac);
  p.jt_row.resize(e.nnz_jac);
  std::vector<int> fill(p.jt_ptr.begin(), p.jt_ptr.end() - 1);
  for (int k = 0; k < e.nnz_jac; ++k) {
    const int q = fill[e.jac_j[k]]++;
    p.jt_ent[q] = k;
    p.jt_row[q] = e.jac_i[k];
  }
  return RPM_OK;
}

}  // namespace rpm

// rpm_ipm.cpp — time-ordered band + border layout of the primal-dual KKT matrix (host side of row f-2, see rpm_ipm.hpp).
#include "rpm_ipm.hpp"

#include <algorithm>
#include <cstdint>
#include <cstdlib>

namespace rpm {

constexpr int IPM_PLAN_W = 16;   // = IPM_W, the factorisation's block width (rpm_ipm_device.hpp)

static int build_ipm_plan_nd(Engine& e, IpmPlan& p, std::string* why);

void ipm_plan_group_hessian(IpmPlan& p) {
  std::vector<std::pair<int, int>> all;   // (slot, COO entry), COO order kept inside a slot
  for (int k = 0; k < int(p.hes_dst.size()); ++k)
    if (p.hes_dst[k] >= 0) all.emplace_back(p.hes_dst[k], k);
  std::stable_sort(all.begin(), all.end(), [](const auto& a, const auto& c) { return a.first < c.first; });
  p.hg_ptr.assign(1, 0);
  p.hg_src.clear();
  p.hg_dst.clear();
  for (size_t i = 0; i < all.size(); ++i) {
    if (i == 0 || all[i].first != all[i - 1].first) {
      if (i) p.hg_ptr.push_back(int(p.hg_src.size()));
      p.hg_dst.push_back(all[i].first);
    }
    p.hg_src.push_back(all[i].second);
  }
  if (!all.empty()) p.hg_ptr.push_back(int(p.hg_src.size()));
}

// Unknowns that must join the border because the Hessian couples them across collocation nodes.  A Lagrangian Hessian of this
// transcription couples variables of ONE node with each other and with the border (t0, tf, final states, parameters); the one
// exception is the reference's linkage Hessian, which indexes the right phase's initial states with the LEFT phase's node count
// (LpHessian.cpp:1150, kept bug-for-bug): when linked phases have different node totals — every mesh hp-Liu refinement produces —
// those entries land on states in the middle of the right phase.  Both endpoints of such an entry (at most nx per linkage) are
// promoted; the band / the interval structure then holds for everything else.
static std::vector<char> promoted_to_border(const Engine& e) {
  std::vector<long long> node(e.n, -1);          // global node index of a variable, -1 = border already
  long long base = 0;
  for (int ip = 0; ip < e.P; ++ip) {
    const PhaseDev& q = e.phd[ip];
    for (int i = 0; i < q.nx; ++i)
      for (int k = 0; k < q.N; ++k) node[q.x_state0 + i * (q.N + 1) + k] = base + k;
    for (int j = 0; j < q.nu; ++j)
      for (int k = 0; k < q.N; ++k) node[q.x_control0 + j * q.N + k] = base + k;
    base += q.N;
  }
  std::vector<char> pr(e.n, 0);
  for (int k = 0; k < e.nnz_h; ++k) {
    const int a = e.hes_i[k], c = e.hes_j[k];
    if (e.xl[a] == e.xu[a] || e.xl[c] == e.xu[c]) continue;
    if (node[a] >= 0 && node[c] >= 0 && node[a] != node[c]) pr[a] = pr[c] = 1;
  }
  return pr;
}

int build_ipm_plan(Engine& e, IpmPlan& p, std::string* why, int nested) {
  if (nested) return build_ipm_plan_nd(e, p, why);
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
  {
    const std::vector<char> pr = promoted_to_border(e);
    for (int u = 0; u < p.n; ++u)
      if (pr[u]) key[u] = -1;
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
  p.Nt_alloc = p.Nt;
  p.max_rows = IPM_PLAN_W + p.b + p.nb;
  ipm_plan_group_hessian(p);
  return RPM_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// nested dissection by mesh interval (rpm_ipm.hpp)
namespace {
struct NdClass {
  std::vector<int> ivl;     // Nt: interval of an interior unknown, -1 otherwise
  std::vector<int> loc;     // Nt: local index inside its interval (interior) or level-2 position (separator / border)
};
size_t factor_lds_of(int b, int nb) {   // = kkt_factor_lds_bytes (rpm_ipm_kernels.hip)
  const size_t W = IPM_PLAN_W;
  return (size_t(b + 24) * W + W * (W + 1) + W * W + W + 2 * size_t(nb) * W + size_t(nb) * (nb + 1) / 2) * sizeof(double);
}
long long sub_at(const KktSubHost& g, int i, int j) { return g.koff + (long long)j * g.CS + (i < g.Nb ? i - j : g.b + 1 + i - g.Nb); }

// destinations with long source lists first (order among them and among the rest unchanged); returns how many
int long_lists_first(std::vector<int>& ptr, std::vector<int>& src, std::vector<int>& dst, int threshold = 32) {
  const int n = int(dst.size());
  std::vector<int> order;
  for (int pass = 0; pass < 2; ++pass)
    for (int i = 0; i < n; ++i)
      if ((ptr[i + 1] - ptr[i] >= threshold) == (pass == 0)) order.push_back(i);
  std::vector<int> nptr(1, 0), nsrc, ndst;
  int n_long = 0;
  for (int i : order) {
    if (ptr[i + 1] - ptr[i] >= threshold) ++n_long;
    nsrc.insert(nsrc.end(), src.begin() + ptr[i], src.begin() + ptr[i + 1]);
    nptr.push_back(int(nsrc.size()));
    ndst.push_back(dst[i]);
  }
  ptr.swap(nptr); src.swap(nsrc); dst.swap(ndst);
  return n_long;
}

// Where a position of the separator system (level-2 position a: separators in time order, then the global border) lives once
// that system is cut into groups (IpmPlan::n_l2 > 0), and the storage slot of a pair of such positions.
struct L3Map {
  const IpmPlan& p;
  int KI;                                   // index of the first level-2 sub-problem in p.subs
  explicit L3Map(const IpmPlan& pl) : p(pl), KI(int(pl.subs.size()) - 1 - pl.n_l2) {}
  int group(int a) const { return std::min(a / p.l3_S, p.l3_G - 1); }
  bool interior(int a) const {
    if (a >= p.l3_Nb2) return false;
    const int g = group(a);
    return g == p.l3_G - 1 || a - g * p.l3_S < p.l3_S - p.l3_w;
  }
  int local(int a) const { return a - group(a) * p.l3_S; }                      // interior index inside its group
  int last_pos(int a) const {                                                   // position in the last level (a not interior)
    if (a >= p.l3_Nb2) return (p.l3_G - 1) * p.l3_w + (a - p.l3_Nb2);
    const int g = group(a);
    return g * p.l3_w + (a - (g * p.l3_S + p.l3_S - p.l3_w));
  }
  int n_prev(int g) const { return g > 0 ? p.l3_w : 0; }
  int n_own(int g) const { return g < p.l3_G - 1 ? p.l3_w : 0; }
  int n_gb(int g) const { return p.l3_gb_ptr[g + 1] - p.l3_gb_ptr[g]; }
  int lborder(int g, int a) const {                                             // local border index of a (not interior) seen from group g, or -1
    if (a >= p.l3_Nb2) {
      const auto first = p.l3_gb.begin() + p.l3_gb_ptr[g], last = p.l3_gb.begin() + p.l3_gb_ptr[g + 1];
      const auto it = std::lower_bound(first, last, a - p.l3_Nb2);
      return it != last && *it == a - p.l3_Nb2 ? n_prev(g) + n_own(g) + int(it - first) : -1;
    }
    const int ga = group(a), idx = a - (ga * p.l3_S + p.l3_S - p.l3_w);
    if (ga == g - 1) return idx;
    if (ga == g) return n_prev(g) + idx;
    return -1;
  }
  int rhs_pos(int a) const {
    if (interior(a)) return p.subs[KI + group(a)].roff + local(a);
    return p.l3_base + last_pos(a);
  }
  long long slot(int a, int c) const {
    const KktSubHost& G3 = p.subs.back();
    const bool ia = interior(a), ic = interior(c);
    if (!ia && !ic) {
      int pa = last_pos(a), pc = last_pos(c);
      if (pa < pc) std::swap(pa, pc);
      if (pa < G3.Nb && pa - pc > G3.b) return -1;
      return sub_at(G3, pa, pc);
    }
    if (ia && ic) {
      if (group(a) != group(c)) return -1;
      int la = local(a), lc = local(c);
      if (la < lc) std::swap(la, lc);
      const KktSubHost& g = p.subs[KI + group(a)];
      if (la - lc > g.b) return -1;
      return sub_at(g, la, lc);
    }
    const int in = ia ? a : c, out = ia ? c : a, g = group(in);
    const int lb = lborder(g, out);
    if (lb < 0) return -1;
    return sub_at(p.subs[KI + g], p.subs[KI + g].Nb + lb, local(in));
  }
};
}  // namespace

static int build_ipm_plan_nd(Engine& e, IpmPlan& p, std::string* why) {
  p = IpmPlan();
  p.nd = 1;
  p.n = e.n;
  p.m = e.m;
  auto fail = [&](const std::string& msg, int code = RPM_E_UNSUPPORTED) {
    if (why) *why = msg;
    return code;
  };
  if (int(e.hes_i.size()) != e.nnz_h || int(e.jac_i.size()) != e.nnz_jac) return fail("the Jacobian / Hessian structure is not built yet", RPM_E_INVALID);
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

  // ---- intervals, and where every unknown lives: interior of an interval (with its node as sort key), separator, border
  struct Ivl { int phase, k0, nk, nx; std::vector<int> interior; int sep0 = 0; bool last = false; };
  std::vector<Ivl> iv;
  std::vector<int> ivl_of(p.Nt, -1), sep_of(p.Nt, -1), sep_state(p.Nt, 0);   // sep_state: a state at an interval's first node
  std::vector<long long> key(p.Nt, -1);
  for (int ip = 0; ip < e.P; ++ip) {
    const PhaseDev& q = e.phd[ip];
    const PhaseHost& ph = e.ph[ip];
    std::vector<int> node_ivl(q.N);
    int r0 = 0;
    const int base = int(iv.size());
    for (int k = 0; k < ph.K; ++k) {
      Ivl I;
      I.phase = ip; I.k0 = r0; I.nk = ph.nk[k]; I.nx = q.nx; I.last = k == ph.K - 1;
      for (int kk = 0; kk < I.nk; ++kk) node_ivl[r0 + kk] = base + k;
      r0 += ph.nk[k];
      iv.push_back(I);
    }
    // Separator of an interval = the states at its first node.  What else lives at that node — controls, slacks and the
    // multipliers of its rows — belongs to the interior but is eliminated LAST there: those rows lean on the separator
    // states (D(0,0) x_0), which are not eliminated before level 2, so eliminated in node order they would meet pivots of
    // -delta_c alone (on the badly scaled climb problem the first Newton step came out 3e-4 off); at the end of the
    // interior they have the interval's other states, D(0,1..N-1), behind them.
    for (int i = 0; i < q.nx; ++i)
      for (int k = 0; k < q.N; ++k) {   // X(N, .) stays in the global border
        const int u = q.x_state0 + i * (q.N + 1) + k, I = node_ivl[k];
        if (k == iv[I].k0) { sep_of[u] = I; sep_state[u] = 1; }
        else { ivl_of[u] = I; key[u] = k; }
      }
    auto place = [&](int u, int k) {
      const int I = node_ivl[k];
      ivl_of[u] = I;
      key[u] = k == iv[I].k0 ? (long long)q.N + 1 : k;   // first node: after every other node of the interval
    };
    for (int j = 0; j < q.nu; ++j)
      for (int k = 0; k < q.N; ++k) place(q.x_control0 + j * q.N + k, k);
    for (int r = 0; r < (q.nx + q.nc) * q.N; ++r) {
      const int row = q.g0 + r, k = r % q.N;
      place(p.nv + row, k);
      if (p.row_slack[row] >= 0) place(p.n + p.row_slack[row], k);
    }
  }
  {
    const std::vector<char> pr = promoted_to_border(e);
    for (int u = 0; u < p.n; ++u)
      if (pr[u]) { ivl_of[u] = -1; sep_of[u] = -1; sep_state[u] = 0; key[u] = -1; }
  }
  const int KI = int(iv.size());
  auto type_of = [&](int u) { return u < p.n ? 0 : (u < p.nv ? 1 : 2); };
  std::vector<int> border;
  std::vector<std::vector<int>> sep(KI);
  for (int u = 0; u < p.Nt; ++u) {
    if (ivl_of[u] >= 0) iv[ivl_of[u]].interior.push_back(u);
    else if (sep_of[u] >= 0) sep[sep_of[u]].push_back(u);     // ascending unknown index: states, controls, slacks, multipliers
    else border.push_back(u);
  }
  for (Ivl& I : iv)
    std::stable_sort(I.interior.begin(), I.interior.end(), [&](int a, int c) {
      if (key[a] != key[c]) return key[a] < key[c];
      return type_of(a) < type_of(c);
    });
  p.nb = int(border.size());

  // ---- positions: [interior_I | border work space of I] for every interval, then level 2 = separators in time order, border
  NdClass C;
  C.ivl = ivl_of;
  C.loc.assign(p.Nt, -1);
  p.pos.assign(p.Nt, 0);
  std::vector<int> l2pos(p.Nt, -1);
  {
    int q2 = 0;
    for (int I = 0; I < KI; ++I) { iv[I].sep0 = q2; for (int u : sep[I]) l2pos[u] = q2++; }
    for (int u : border) l2pos[u] = q2++;
  }
  const int Nb2 = [&] { int s2 = 0; for (auto& v : sep) s2 += int(v.size()); return s2; }();
  // the states among a separator's unknowns come first (ascending unknown index, states have the smallest indices of a phase)
  std::vector<int> nstate0(KI, 0);
  for (int I = 0; I < KI; ++I)
    for (int u : sep[I]) nstate0[I] += sep_state[u];
  // the unknowns of the global border that an interval's interior has entries with (its phase's t0, tf, the phase's final states
  // for its last interval, ...): only these get rows in the interval's block — the others' rows of L would be zeros, carried
  // through every panel, corner update, gather and substitution.  Engine option ipm_local_border 0: every interval carries them all.
  std::vector<std::vector<int>> gb(KI);
  if (e.opt_ipm_local_border) {
    auto touch = [&](int ua, int uc) {
      const int Ia = ivl_of[ua], Ic = ivl_of[uc];
      if ((Ia >= 0) == (Ic >= 0)) return;
      const int I = Ia >= 0 ? Ia : Ic, u = Ia >= 0 ? uc : ua;
      if (sep_of[u] < 0) gb[I].push_back(l2pos[u] - Nb2);
    };
    for (int k = 0; k < e.nnz_jac; ++k)
      if (!p.fixed[e.jac_j[k]]) touch(p.nv + e.jac_i[k], e.jac_j[k]);
    for (int k = 0; k < e.nnz_h; ++k)
      if (!p.fixed[e.hes_i[k]] && !p.fixed[e.hes_j[k]]) touch(e.hes_i[k], e.hes_j[k]);
    for (int s = 0; s < p.ns; ++s) touch(p.nv + p.slack_row[s], p.n + s);
    for (auto& v : gb) {
      std::sort(v.begin(), v.end());
      v.erase(std::unique(v.begin(), v.end()), v.end());
    }
  } else {
    for (auto& v : gb)
      for (int j = 0; j < p.nb; ++j) v.push_back(j);
  }
  auto gb_index = [&](int I, int j) -> int {   // local number of global-border unknown j in interval I, or -1
    const auto it = std::lower_bound(gb[I].begin(), gb[I].end(), j);
    return it != gb[I].end() && *it == j ? int(it - gb[I].begin()) : -1;
  };
  std::vector<int> base(KI), nI(KI), nbL(KI);
  int cur = 0;
  for (int I = 0; I < KI; ++I) {
    base[I] = cur;
    nI[I] = int(iv[I].interior.size());
    nbL[I] = int(sep[I].size()) + (iv[I].last ? 0 : nstate0[I + 1]) + int(gb[I].size());   // own separator, the NEXT interval's first-node states, its part of the border
    for (int q = 0; q < nI[I]; ++q) { const int u = iv[I].interior[q]; C.loc[u] = q; p.pos[u] = cur + q; }
    cur += nI[I] + nbL[I];
  }
  const int l2base = cur;
  for (int u = 0; u < p.Nt; ++u)
    if (ivl_of[u] < 0) C.loc[u] = l2pos[u];      // p.pos of these follows once the separator system's own layout is known
  p.Nb = Nb2;   // reported by rpm_ipm_get_info as the banded part of level 2

  // local border index of a separator / border unknown seen from interval I, or -1
  auto lborder = [&](int I, int u) -> int {
    if (sep_of[u] == I) return l2pos[u] - iv[I].sep0;
    if (!iv[I].last && sep_of[u] == I + 1) return sep_state[u] ? int(sep[I].size()) + (l2pos[u] - iv[I + 1].sep0) : -1;
    if (sep_of[u] < 0 && ivl_of[u] < 0) {
      const int j = gb_index(I, l2pos[u] - Nb2);
      return j < 0 ? -1 : int(sep[I].size()) + (iv[I].last ? 0 : nstate0[I + 1]) + j;
    }
    return -1;
  };

  // ---- half bandwidths from the entries
  std::vector<int> bI(KI, 0);
  int b2 = 0;
  bool bad_pair = false;
  auto reach = [&](int ua, int uc) {
    const int Ia = ivl_of[ua], Ic = ivl_of[uc];
    if (Ia >= 0 && Ic >= 0) {
      if (Ia != Ic) { bad_pair = true; return; }
      bI[Ia] = std::max(bI[Ia], std::abs(C.loc[ua] - C.loc[uc]));
    } else if (Ia >= 0 || Ic >= 0) {
      const int I = Ia >= 0 ? Ia : Ic, u = Ia >= 0 ? uc : ua;
      if (lborder(I, u) < 0) bad_pair = true;
    } else if (l2pos[ua] < Nb2 && l2pos[uc] < Nb2) {
      b2 = std::max(b2, std::abs(l2pos[ua] - l2pos[uc]));
    }
  };
  for (int k = 0; k < e.nnz_jac; ++k)
    if (!p.fixed[e.jac_j[k]]) reach(p.nv + e.jac_i[k], e.jac_j[k]);
  for (int k = 0; k < e.nnz_h; ++k)
    if (!p.fixed[e.hes_i[k]] && !p.fixed[e.hes_j[k]]) reach(e.hes_i[k], e.hes_j[k]);
  for (int s = 0; s < p.ns; ++s) reach(p.nv + p.slack_row[s], p.n + s);
  if (bad_pair) return fail("nested dissection: an entry couples the interiors of two mesh intervals, or an interior to a separator that is not its own");
  for (int I = 0; I < KI; ++I)   // the Schur complements couple an interval's separator with the next one's states
    if (!iv[I].last) b2 = std::max(b2, int(sep[I].size()) + nstate0[I + 1] - 1);

  // ---- third level?  (the separator system of a long mesh is cut again, at the matrix level: rpm_ipm.hpp)
  {
    const int w = std::max(b2, 1);
    int S = e.opt_ipm_nested_group;
    if (S == 0 && Nb2 >= 512) {                  // automatic: group size ~ sqrt(Nb2 * w), the minimum of (S - w) + (Nb2 / S) w block columns
      S = int(std::sqrt(double(Nb2) * w));
      S = (S + 15) / 16 * 16;
    }
    if (S > 0) {
      S = std::max(S, 3 * w);
      const int G = Nb2 / S;                     // the last group takes the remainder
      if (G >= 2) { p.n_l2 = G; p.l3_S = S; p.l3_w = w; p.l3_G = G; p.l3_Nb2 = Nb2; }
    }
  }
  if (p.n_l2) {
    // the global-border unknowns a group's interior has entries with — entries of the KKT matrix itself, and of the intervals' Schur
    // complements (every pair of an interval's local border): only these get rows in the group's block, as with the intervals
    std::vector<std::vector<int>> gb2(size_t(p.l3_G));
    if (e.opt_ipm_local_border) {
      const L3Map M(p);
      auto touch = [&](int a, int c) {
        if (a < c) std::swap(a, c);
        if (a >= Nb2 && c < Nb2 && M.interior(c)) gb2[size_t(M.group(c))].push_back(a - Nb2);
      };
      auto entry = [&](int ua, int uc) {
        if (ivl_of[ua] < 0 && ivl_of[uc] < 0) touch(l2pos[ua], l2pos[uc]);
      };
      for (int k = 0; k < e.nnz_jac; ++k)
        if (!p.fixed[e.jac_j[k]]) entry(p.nv + e.jac_i[k], e.jac_j[k]);
      for (int k = 0; k < e.nnz_h; ++k)
        if (!p.fixed[e.hes_i[k]] && !p.fixed[e.hes_j[k]]) entry(e.hes_i[k], e.hes_j[k]);
      for (int s = 0; s < p.ns; ++s) entry(p.nv + p.slack_row[s], p.n + s);
      for (int I = 0; I < KI; ++I) {
        std::vector<int> seps;                       // the separator positions in I's local border
        for (int u : sep[I]) seps.push_back(l2pos[u]);
        if (!iv[I].last)
          for (int u : sep[I + 1])
            if (sep_state[u]) seps.push_back(l2pos[u]);
        for (int a : seps)
          for (int j : gb[I]) touch(a, Nb2 + j);
      }
      for (auto& v : gb2) {
        std::sort(v.begin(), v.end());
        v.erase(std::unique(v.begin(), v.end()), v.end());
      }
    } else {
      for (auto& v : gb2)
        for (int j = 0; j < p.nb; ++j) v.push_back(j);
    }
    p.l3_gb_ptr.assign(1, 0);
    for (const auto& v : gb2) {
      p.l3_gb.insert(p.l3_gb.end(), v.begin(), v.end());
      p.l3_gb_ptr.push_back(int(p.l3_gb.size()));
    }
  }

  // ---- sub-problem geometry and storage
  long long koff = 0;
  p.subs.clear();
  for (int I = 0; I < KI; ++I) {
    KktSubHost g;
    g.Nb = nI[I];
    g.nb = nbL[I];
    g.Nt = g.Nb + g.nb;
    g.b = std::max(bI[I], IPM_PLAN_W);
    if (g.b > g.Nb - 1) g.b = std::max(g.Nb - 1, 0);
    g.CS = g.b + 1 + g.nb;
    g.CS += g.CS & 1;
    g.koff = koff;
    g.roff = base[I];
    koff += (long long)g.Nt * g.CS;
    p.subs.push_back(g);
  }
  {
    int roff = l2base;
    auto push = [&](int Nb_, int nb_, int b_) {
      KktSubHost g;
      g.Nb = Nb_;
      g.nb = nb_;
      g.Nt = Nb_ + nb_;
      g.b = std::max(b_, IPM_PLAN_W);
      if (g.b > g.Nb - 1) g.b = std::max(g.Nb - 1, 0);
      g.CS = g.b + 1 + g.nb;
      g.CS += g.CS & 1;
      g.koff = koff;
      g.roff = roff;
      koff += (long long)g.Nt * g.CS;
      roff += g.Nt;
      p.subs.push_back(g);
    };
    if (p.n_l2) {
      for (int g = 0; g < p.l3_G; ++g) {
        const int n_g = g < p.l3_G - 1 ? p.l3_S - p.l3_w : Nb2 - g * p.l3_S;
        push(n_g, (g > 0 ? p.l3_w : 0) + (g < p.l3_G - 1 ? p.l3_w : 0) + (p.l3_gb_ptr[size_t(g) + 1] - p.l3_gb_ptr[size_t(g)]), b2);
      }
      p.l3_base = roff;
      push((p.l3_G - 1) * p.l3_w, p.nb, 2 * p.l3_w - 1);
    } else {
      push(Nb2, p.nb, b2);
    }
    p.b = p.subs.back().b;
    p.CS = p.subs.back().CS;
    p.Nt_alloc = roff;
  }
  const L3Map M3(p);
  auto l2_rhs = [&](int a) { return p.n_l2 ? M3.rhs_pos(a) : l2base + a; };
  auto l2_slot = [&](int a, int c) -> long long {
    if (p.n_l2) return M3.slot(a, c);
    if (a < c) std::swap(a, c);
    const KktSubHost& G2 = p.subs.back();
    if (a < Nb2 && a - c > G2.b) return -1;
    return sub_at(G2, a, c);
  };
  for (int u = 0; u < p.Nt; ++u)
    if (ivl_of[u] < 0) p.pos[u] = l2_rhs(l2pos[u]);
  p.storage_nd = koff;
  if (p.storage_nd > INT32_MAX) return fail("KKT storage of one instance exceeds 2^31 doubles");
  p.max_rows = 0;
  p.max_factor_lds = 0;
  for (const KktSubHost& g : p.subs) {
    p.max_rows = std::max(p.max_rows, IPM_PLAN_W + g.b + g.nb);
    p.max_factor_lds = std::max(p.max_factor_lds, factor_lds_of(g.b, g.nb));
  }

  // ---- storage slot of every entry
  auto dst = [&](int ua, int uc) -> long long {
    const int Ia = ivl_of[ua], Ic = ivl_of[uc];
    if (Ia >= 0 && Ic >= 0) {
      if (Ia != Ic) return -1;
      int a = C.loc[ua], c = C.loc[uc];
      if (a < c) std::swap(a, c);
      if (a - c > p.subs[Ia].b) return -1;
      return sub_at(p.subs[Ia], a, c);
    }
    if (Ia >= 0 || Ic >= 0) {
      const int I = Ia >= 0 ? Ia : Ic, ui = Ia >= 0 ? ua : uc, uo = Ia >= 0 ? uc : ua;
      const int lb = lborder(I, uo);
      if (lb < 0) return -1;
      return sub_at(p.subs[I], p.subs[I].Nb + lb, C.loc[ui]);
    }
    return l2_slot(l2pos[ua], l2pos[uc]);
  };
  bool missing = false;
  auto dst_i = [&](int ua, int uc) -> int {
    const long long o = dst(ua, uc);
    if (o < 0) missing = true;
    return int(o);
  };
  p.jac_dst.assign(e.nnz_jac, -1);
  p.hes_dst.assign(e.nnz_h, -1);
  for (int k = 0; k < e.nnz_jac; ++k)
    if (!p.fixed[e.jac_j[k]]) p.jac_dst[k] = dst_i(p.nv + e.jac_i[k], e.jac_j[k]);
  for (int k = 0; k < e.nnz_h; ++k)
    if (!p.fixed[e.hes_i[k]] && !p.fixed[e.hes_j[k]]) p.hes_dst[k] = dst_i(e.hes_i[k], e.hes_j[k]);
  p.diag_dst.resize(p.Nt);
  for (int u = 0; u < p.Nt; ++u) p.diag_dst[u] = dst_i(u, u);
  p.slk_dst.resize(p.ns);
  for (int s = 0; s < p.ns; ++s) p.slk_dst[s] = dst_i(p.nv + p.slack_row[s], p.n + s);
  if (missing) return fail("nested dissection: an entry of the KKT matrix has no slot in the interval / separator layout");

  // ---- corner gather (level-1 Schur complements -> level 2), right-hand-side gather and solution scatter, in interval order
  {
    std::vector<std::vector<int>> src_of;          // per distinct level-2 slot
    std::vector<long long> slot_key;
    std::vector<std::pair<long long, int>> all;    // (level-2 offset, level-1 offset), generated in interval order
    std::vector<std::pair<int, int>> rall;         // (level-2 rhs position, work-space position)
    for (int I = 0; I < KI; ++I) {
      const KktSubHost& g = p.subs[I];
      std::vector<int> l2_of(g.nb);                // level-2 position of local border index
      int q = 0;
      for (int u : sep[I]) l2_of[q++] = l2pos[u];
      if (!iv[I].last)
        for (int u : sep[I + 1])
          if (sep_state[u]) l2_of[q++] = l2pos[u];
      for (int j : gb[I]) l2_of[q++] = Nb2 + j;
      for (int r = 0; r < g.nb; ++r) {
        for (int c = 0; c <= r; ++c) {
          const long long o = l2_slot(l2_of[r], l2_of[c]);
          if (o < 0) return fail("nested dissection: a Schur complement entry of an interval has no slot in the separator system's layout");
          all.emplace_back(o, int(sub_at(g, g.Nb + r, g.Nb + c)));
        }
        rall.emplace_back(l2_rhs(l2_of[r]), g.roff + g.Nb + r);
        p.gap_pos.push_back(g.roff + g.Nb + r);
        p.rs_dst.push_back(g.roff + g.Nb + r);
        p.rs_src.push_back(l2_rhs(l2_of[r]));
      }
    }
    std::stable_sort(all.begin(), all.end(), [](const auto& a, const auto& c) { return a.first < c.first; });
    p.cg_ptr.push_back(0);
    for (size_t i = 0; i < all.size(); ++i) {
      if (i == 0 || all[i].first != all[i - 1].first) {
        if (i) p.cg_ptr.push_back(int(p.cg_src.size()));
        p.cg_dst.push_back(int(all[i].first));
      }
      p.cg_src.push_back(all[i].second);
    }
    p.cg_ptr.push_back(int(p.cg_src.size()));
    if (all.empty()) p.cg_ptr.assign(1, 0);
    p.n_cg_long = long_lists_first(p.cg_ptr, p.cg_src, p.cg_dst);
    std::stable_sort(rall.begin(), rall.end(), [](const auto& a, const auto& c) { return a.first < c.first; });
    p.rg_ptr.push_back(0);
    for (size_t i = 0; i < rall.size(); ++i) {
      if (i == 0 || rall[i].first != rall[i - 1].first) {
        if (i) p.rg_ptr.push_back(int(p.rg_src.size()));
        p.rg_dst.push_back(rall[i].first);
      }
      p.rg_src.push_back(rall[i].second);
    }
    p.rg_ptr.push_back(int(p.rg_src.size()));
    if (rall.empty()) p.rg_ptr.assign(1, 0);
  }
  if (p.n_l2) {   // second stage: the group sub-problems' Schur complements -> the last level, in group order
    const KktSubHost& G3 = p.subs.back();
    std::vector<std::pair<long long, int>> all;
    std::vector<std::pair<int, int>> rall;
    for (int g = 0; g < p.l3_G; ++g) {
      const KktSubHost& q = p.subs[KI + g];
      std::vector<int> l3_of;                      // last-level position of local border index
      if (g > 0) for (int j = 0; j < p.l3_w; ++j) l3_of.push_back((g - 1) * p.l3_w + j);
      if (g < p.l3_G - 1) for (int j = 0; j < p.l3_w; ++j) l3_of.push_back(g * p.l3_w + j);
      for (int q2 = p.l3_gb_ptr[size_t(g)]; q2 < p.l3_gb_ptr[size_t(g) + 1]; ++q2) l3_of.push_back(G3.Nb + p.l3_gb[size_t(q2)]);
      for (int r = 0; r < q.nb; ++r) {
        for (int c = 0; c <= r; ++c) {
          if (l3_of[r] < G3.Nb && l3_of[r] - l3_of[c] > G3.b) return fail("nested dissection: third-level band too narrow");
          all.emplace_back(sub_at(G3, l3_of[r], l3_of[c]), int(sub_at(q, q.Nb + r, q.Nb + c)));
        }
        rall.emplace_back(p.l3_base + l3_of[r], q.roff + q.Nb + r);
        p.gap_pos.push_back(q.roff + q.Nb + r);
        p.rs2_dst.push_back(q.roff + q.Nb + r);
        p.rs2_src.push_back(p.l3_base + l3_of[r]);
      }
    }
    std::stable_sort(all.begin(), all.end(), [](const auto& a, const auto& c) { return a.first < c.first; });
    p.cg2_ptr.push_back(0);
    for (size_t i = 0; i < all.size(); ++i) {
      if (i == 0 || all[i].first != all[i - 1].first) {
        if (i) p.cg2_ptr.push_back(int(p.cg2_src.size()));
        p.cg2_dst.push_back(int(all[i].first));
      }
      p.cg2_src.push_back(all[i].second);
    }
    p.cg2_ptr.push_back(int(p.cg2_src.size()));
    p.n_cg2_long = long_lists_first(p.cg2_ptr, p.cg2_src, p.cg2_dst);
    std::stable_sort(rall.begin(), rall.end(), [](const auto& a, const auto& c) { return a.first < c.first; });
    p.rg2_ptr.push_back(0);
    for (size_t i = 0; i < rall.size(); ++i) {
      if (i == 0 || rall[i].first != rall[i - 1].first) {
        if (i) p.rg2_ptr.push_back(int(p.rg2_src.size()));
        p.rg2_dst.push_back(rall[i].first);
      }
      p.rg2_src.push_back(rall[i].second);
    }
    p.rg2_ptr.push_back(int(p.rg2_src.size()));
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
  ipm_plan_group_hessian(p);
  // keep what ipm_plan_offset needs
  p.nd_ivl = ivl_of;
  p.nd_loc = C.loc;
  p.nd_sep = sep_of;
  p.nd_sep0.resize(KI);
  p.nd_nsep.resize(KI);
  p.nd_last.resize(KI);
  p.nd_nstate0 = nstate0;
  p.nd_sep_state = sep_state;
  p.nd_gb_ptr.assign(1, 0);
  for (int I = 0; I < KI; ++I) {
    p.nd_gb.insert(p.nd_gb.end(), gb[I].begin(), gb[I].end());
    p.nd_gb_ptr.push_back(int(p.nd_gb.size()));
  }
  for (int I = 0; I < KI; ++I) { p.nd_sep0[I] = iv[I].sep0; p.nd_nsep[I] = int(sep[I].size()); p.nd_last[I] = iv[I].last ? 1 : 0; }
  return RPM_OK;
}

long long ipm_plan_offset(const IpmPlan& p, int ua, int uc) {
  if (!p.nd) {
    int a = p.pos[ua], c = p.pos[uc];
    if (a < c) std::swap(a, c);
    if (a < p.Nb && a - c > p.b) return -1;
    return p.at(a, c);
  }
  const KktSubHost& G2 = p.subs.back();
  const int Nb2 = p.n_l2 ? p.l3_Nb2 : G2.Nb;
  auto lborder = [&](int I, int u) -> int {
    if (p.nd_sep[u] == I) return p.nd_loc[u] - p.nd_sep0[I];
    if (!p.nd_last[I] && p.nd_sep[u] == I + 1) return p.nd_sep_state[u] ? p.nd_nsep[I] + (p.nd_loc[u] - p.nd_sep0[I + 1]) : -1;
    if (p.nd_sep[u] < 0 && p.nd_ivl[u] < 0) {
      const auto first = p.nd_gb.begin() + p.nd_gb_ptr[I], last = p.nd_gb.begin() + p.nd_gb_ptr[I + 1];
      const auto it = std::lower_bound(first, last, p.nd_loc[u] - Nb2);
      if (it == last || *it != p.nd_loc[u] - Nb2) return -1;
      return p.nd_nsep[I] + (p.nd_last[I] ? 0 : p.nd_nstate0[I + 1]) + int(it - first);
    }
    return -1;
  };
  const int Ia = p.nd_ivl[ua], Ic = p.nd_ivl[uc];
  if (Ia >= 0 && Ic >= 0) {
    if (Ia != Ic) return -1;
    int a = p.nd_loc[ua], c = p.nd_loc[uc];
    if (a < c) std::swap(a, c);
    if (a - c > p.subs[Ia].b) return -1;
    return sub_at(p.subs[Ia], a, c);
  }
  if (Ia >= 0 || Ic >= 0) {
    const int I = Ia >= 0 ? Ia : Ic, ui = Ia >= 0 ? ua : uc, uo = Ia >= 0 ? uc : ua;
    const int lb = lborder(I, uo);
    if (lb < 0) return -1;
    return sub_at(p.subs[I], p.subs[I].Nb + lb, p.nd_loc[ui]);
  }
  int a = p.nd_loc[ua], c = p.nd_loc[uc];
  if (p.n_l2) return L3Map(p).slot(a, c);
  if (a < c) std::swap(a, c);
  if (a < Nb2 && a - c > G2.b) return -1;
  return sub_at(G2, a, c);
}

}  // namespace rpm

// rpm_ipm_solver.hip — row f-2, host side: the interior-point loop over the batched kernels of rpm_ipm_kernels.hip (per
// iteration three counters come back from the device, nothing else) and the rpm_ipm_* entry points of the C ABI.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <unordered_map>
#include <vector>

#include "rpm_device_internal.hpp"
#include "rpm_ipm_device.hpp"

// =================================================================================================== host side + ABI
using namespace rpm;


struct rpm_ipm {
  rpm_engine* eng = nullptr;
  IpmPlan plan;
  IpmDev D{};
  std::vector<void*> allocs;
  int* h_cnt = nullptr;           // page-locked mirror of D.cnt
  size_t factor_lds = 0;
  size_t l1_dense_lds = 0;    // LDS of kkt_factor_dense_kernel when every level-1 sub-problem fits its register tiles, else 0
  size_t l2_dense_lds = 0, last_dense_lds = 0;   // the same for the groups of separators and for the last level
  int factor_mt = IPM_MT;
  std::string err;
  std::vector<IpmInst> h_inst;
  int total_factorizations = 0, total_iterations = 0, total_trials = 0, total_soc = 0;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // around the factorisation and the substitution of an iteration
  double factor_ms = 0.0, solve_ms = 0.0;
  bool solve_pending = false;
  bool attached = false;
  bool lbfgs = false;            // hessian-approximation = limited-memory (rpm_ipm_lbfgs.hip)
  int lb_iterations = 0;         // iterations of the running solve: an upper bound of the pairs any instance holds
  ~rpm_ipm() {
    if (attached && eng && eng->e.ipm_attached > 0) eng->e.ipm_attached -= 1;
    for (void* p : allocs) (void)hipFree(p);
    if (h_cnt) (void)hipHostFree(h_cnt);
    for (hipEvent_t e2 : ev)
      if (e2) (void)hipEventDestroy(e2);
  }
};

#define IPM_TRY(h, call)                                                      \
  do {                                                                        \
    hipError_t _s = (call);                                                   \
    if (_s != hipSuccess) {                                                   \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(_s);          \
      return RPM_E_DEVICE;                                                    \
    }                                                                         \
  } while (0)

namespace {
template <class T>
int ipm_alloc(rpm_ipm* h, T** dst, size_t count, const T* src = nullptr) {
  void* p = nullptr;
  IPM_TRY(h, hipMalloc(&p, (count ? count : 1) * sizeof(T)));
  h->allocs.push_back(p);
  *dst = static_cast<T*>(p);
  if (src && count) IPM_TRY(h, hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice));
  return RPM_OK;
}
template <class T>
int ipm_alloc_c(rpm_ipm* h, const T** dst, const std::vector<T>& src) {
  T* p = nullptr;
  int rc = ipm_alloc(h, &p, src.size(), src.data());
  *dst = p;
  return rc;
}

int fetch_counts(rpm_ipm* h, hipStream_t st) {
  IPM_TRY(h, hipMemcpyAsync(h->h_cnt, h->D.cnt, 4 * sizeof(int), hipMemcpyDeviceToHost, st));
  IPM_TRY(h, hipStreamSynchronize(st));
  return RPM_OK;
}
int launch_check(rpm_ipm* h, const char* what) {
  hipError_t s = hipGetLastError();
  if (s != hipSuccess) {
    h->err = std::string(what) + ": " + hipGetErrorString(s);
    return RPM_E_DEVICE;
  }
  return RPM_OK;
}
int factor_and_solve_launch(rpm_ipm* h, hipStream_t st, bool factor, bool solve, int check_status, int forward_done = 0) {
  if (factor) kkt_launch_factor(h->D, h->factor_mt, h->factor_lds, st);
  if (solve) kkt_launch_solve(h->D, check_status, st, forward_done);
  return launch_check(h, "kkt kernels");
}
}  // namespace

extern "C" {

int rpm_ipm_create(rpm_engine* eng, rpm_ipm** out) {
  if (!eng || !out) return RPM_E_INVALID;
  *out = nullptr;
  Engine& e = eng->e;
  // hessian-approximation: exact = lpopc's finite-difference Hessian (rpm_eval_h); limited-memory = the reference's default
  // (Core/LpNLPWrapper.hpp:71): Ipopt's limited-memory BFGS, restated in rpm_ipm_lbfgs.hip
  const bool lbfgs = e.hessian_mode != RPM_HESSIAN_EXACT;
  if (e.shard_world > 1) {
    e.err = "rpm_ipm_create: interval-sharded engines are not supported (shard instances across ranks instead)";
    return RPM_E_UNSUPPORTED;
  }
  if (!e.dev) {
    int rc = device_init(e, 0);
    if (rc) return rc;
  }
  int rc = lbfgs ? RPM_OK : ensure_hessian(e);
  if (rc) return rc;
  rpm_ipm* h = new (std::nothrow) rpm_ipm;
  if (!h) return RPM_E_INVALID;
  h->eng = eng;
  h->lbfgs = lbfgs;
  e.ipm_attached += 1;   // freezes the engine's instance strides (rpm_set_option "instance_align"); released by ~rpm_ipm
  h->attached = true;
  std::string why;
  rc = build_ipm_plan(e, h->plan, &why, e.opt_ipm_nested != 0);
  // automatic: no interval structure to dissect, or sub-problems the factorisation kernel cannot hold (rows per block column, LDS:
  // e.g. a border grown by promoted unknowns on top of the intervals' separators) -> one band
  if (e.opt_ipm_nested == -1 && (rc || h->plan.max_rows > 4 * IPM_MT * 16 || kkt_factor_lds_bytes(h->plan) > 150 * 1024))
    rc = build_ipm_plan(e, h->plan, &why, 0);
  if (rc) {
    e.err = "rpm_ipm_create: " + why;
    delete h;
    return rc;
  }
  const IpmPlan& p = h->plan;
  IpmDev& D = h->D;
  const size_t B = size_t(e.n_instances);
  D.B = int(B); D.n = p.n; D.m = p.m; D.ns = p.ns; D.nv = p.nv; D.Nt = p.Nt_alloc; D.Nb = p.Nb; D.nb = p.nb; D.b = p.b; D.CS = p.CS;
  D.nnz_jac = e.nnz_jac; D.nnz_h = e.nnz_h;
  D.sg = e.stride_g(); D.sv = e.stride_values(); D.kstride = p.storage();
  auto fail = [&](int code) { e.err = "rpm_ipm_create: " + h->err; delete h; return code; };
#define A_(call) do { int _r = (call); if (_r) return fail(_r); } while (0)
  A_(ipm_alloc_c(h, &D.pos, p.pos)); A_(ipm_alloc_c(h, &D.row_slack, p.row_slack)); A_(ipm_alloc_c(h, &D.slack_row, p.slack_row));
  A_(ipm_alloc_c(h, &D.jac_dst, p.jac_dst)); A_(ipm_alloc_c(h, &D.hes_dst, p.hes_dst));
  A_(ipm_alloc_c(h, &D.diag_dst, p.diag_dst)); A_(ipm_alloc_c(h, &D.slk_dst, p.slk_dst)); A_(ipm_alloc_c(h, &D.jt_ptr, p.jt_ptr));
  A_(ipm_alloc_c(h, &D.jt_ent, p.jt_ent)); A_(ipm_alloc_c(h, &D.jt_row, p.jt_row));
  A_(ipm_alloc_c(h, &D.hg_ptr, p.hg_ptr)); A_(ipm_alloc_c(h, &D.hg_src, p.hg_src)); A_(ipm_alloc_c(h, &D.hg_dst, p.hg_dst));
  D.n_hg = int(p.hg_dst.size());
  struct Ent { int dst, ki, hg; };
  std::vector<Ent> ents;      // the structural slots in ascending order
  {   // ... for the one-pass fill (ipm_fill_kernel)
    ents.reserve(p.hg_dst.size() + p.jac_dst.size() + p.slk_dst.size() + p.diag_dst.size());
    std::unordered_map<int, int> var_of_slot;
    for (int i = 0; i < p.nv; ++i) var_of_slot.emplace(p.diag_dst[size_t(i)], i);
    std::vector<int> hg_of_var(size_t(p.nv), -1);
    bool ok = p.hg_dst.size() < (1u << 28) && p.jac_dst.size() < (1u << 28) && p.diag_dst.size() < (1u << 28) && p.storage() < (1ll << 31);
    for (size_t i = 0; i < p.hg_dst.size(); ++i) {
      auto it = var_of_slot.find(p.hg_dst[i]);
      if (it != var_of_slot.end()) hg_of_var[size_t(it->second)] = int(i);
      else ents.push_back(Ent{p.hg_dst[i], (0 << 28) | int(i), -1});
    }
    for (size_t k = 0; k < p.jac_dst.size(); ++k)
      if (p.jac_dst[k] >= 0) ents.push_back(Ent{p.jac_dst[k], (1 << 28) | int(k), -1});
    for (size_t s2 = 0; s2 < p.slk_dst.size(); ++s2) ents.push_back(Ent{p.slk_dst[s2], (2 << 28) | int(s2), -1});
    for (int i = 0; i < p.nv; ++i) ents.push_back(Ent{p.diag_dst[size_t(i)], (3 << 28) | i, hg_of_var[size_t(i)]});
    for (int r = 0; r < p.m; ++r) ents.push_back(Ent{p.diag_dst[size_t(p.nv + r)], (4 << 28) | r, -1});
    std::sort(ents.begin(), ents.end(), [](const Ent& a, const Ent& b) { return a.dst < b.dst; });
    for (size_t i = 1; i < ents.size() && ok; ++i) ok = ents[i].dst != ents[i - 1].dst;   // two writers of one slot: keep the two-kernel path
    for (const Ent& en : ents) ok = ok && en.dst >= 0 && en.dst < p.storage();
    D.as_nchunk = 0;
    D.as_dst = D.as_ki = D.as_hg = D.as_ptr = nullptr;
    if (ok && !getenv("RPM_IPM_TWO_PASS_FILL")) {
      const int nchunk = int((p.storage() + IPM_FILL_CHUNK - 1) / IPM_FILL_CHUNK);
      std::vector<int> a_dst(ents.size()), a_ki(ents.size()), a_hg(ents.size()), a_ptr(size_t(nchunk) + 1, 0);
      size_t e2 = 0;
      for (int c = 0; c <= nchunk; ++c) {
        while (e2 < ents.size() && ents[e2].dst < (long long)c * IPM_FILL_CHUNK) ++e2;
        a_ptr[size_t(c)] = int(e2);
      }
      a_ptr[size_t(nchunk)] = int(ents.size());
      for (size_t i = 0; i < ents.size(); ++i) { a_dst[i] = ents[i].dst; a_ki[i] = ents[i].ki; a_hg[i] = ents[i].hg; }
      A_(ipm_alloc_c(h, &D.as_dst, a_dst)); A_(ipm_alloc_c(h, &D.as_ki, a_ki)); A_(ipm_alloc_c(h, &D.as_hg, a_hg)); A_(ipm_alloc_c(h, &D.as_ptr, a_ptr));
      D.as_nchunk = nchunk;
    }
  }
  {
    std::vector<int> long_cols;
    for (int i = 0; i < p.n; ++i)
      if (p.jt_ptr[i + 1] - p.jt_ptr[i] > 256) long_cols.push_back(i);   // = IPM_LONG_COLUMN (rpm_ipm_kernels.hip)
    D.n_long = int(long_cols.size());
    A_(ipm_alloc_c(h, &D.long_cols, long_cols));
  }
  A_(ipm_alloc_c(h, &D.gl, e.gl)); A_(ipm_alloc_c(h, &D.gu, e.gu));
  A_(ipm_alloc(h, &D.v, B * p.nv)); A_(ipm_alloc(h, &D.vl, B * p.nv)); A_(ipm_alloc(h, &D.vu, B * p.nv));
  A_(ipm_alloc(h, &D.zL, B * p.nv)); A_(ipm_alloc(h, &D.zU, B * p.nv)); A_(ipm_alloc(h, &D.lam, B * p.m));
  A_(ipm_alloc(h, &D.dv, B * p.nv)); A_(ipm_alloc(h, &D.dlam, B * p.m)); A_(ipm_alloc(h, &D.dzL, B * p.nv));
  A_(ipm_alloc(h, &D.dzU, B * p.nv)); A_(ipm_alloc(h, &D.glag, B * p.nv)); A_(ipm_alloc(h, &D.c, B * p.m));
  A_(ipm_alloc(h, &D.rhs, B * size_t(p.Nt_alloc))); A_(ipm_alloc(h, &D.K, B * size_t(p.storage()))); A_(ipm_alloc(h, &D.filt, B * 2 * IPM_FMAX));
  A_(ipm_alloc(h, &D.xe, B * p.n)); A_(ipm_alloc(h, &D.xt, B * p.n)); A_(ipm_alloc(h, &D.grad, B * p.n));
  A_(ipm_alloc(h, &D.g, B * size_t(D.sg))); A_(ipm_alloc(h, &D.jac, B * size_t(D.sv))); A_(ipm_alloc(h, &D.hess, B * size_t(e.nnz_h)));
  A_(ipm_alloc(h, &D.obj, B)); A_(ipm_alloc(h, &D.gt, B * size_t(D.sg))); A_(ipm_alloc(h, &D.objt, B));
  A_(ipm_alloc(h, &D.inst, B)); A_(ipm_alloc(h, &D.cnt, size_t(4)));
  A_(ipm_alloc(h, &D.vR, B * p.nv)); A_(ipm_alloc(h, &D.dr2, B * p.nv));
  A_(ipm_alloc(h, &D.vl0, B * p.nv)); A_(ipm_alloc(h, &D.vu0, B * p.nv));
  D.rhs_mult = 1;
  D.lb_on = lbfgs ? 1 : 0;
  D.lb_S = D.lb_Y = D.lb_xprev = D.lb_gold = D.lb_small = D.lb_Z = D.lb_part = nullptr;
  if (lbfgs) {
    A_(ipm_alloc(h, &D.lb_S, B * IPM_LB_H * p.n)); A_(ipm_alloc(h, &D.lb_Y, B * IPM_LB_H * p.n));
    A_(ipm_alloc(h, &D.lb_xprev, B * p.n)); A_(ipm_alloc(h, &D.lb_gold, B * p.nv));
    A_(ipm_alloc(h, &D.lb_small, B * IPM_LB_SMALL)); A_(ipm_alloc(h, &D.lb_part, B * IPM_LB_PART));
    A_(ipm_alloc(h, &D.lb_Z, size_t(2 * IPM_LB_H) * B * size_t(p.Nt_alloc)));
  }
  {   // nlp_scaling (off until the option asks for it; the arrays are small)
    const size_t Bm = B * size_t(std::max(p.m, 1));
    D.scal_on = 0;
    A_(ipm_alloc(h, &D.sc, Bm)); A_(ipm_alloc(h, &D.sf, B)); A_(ipm_alloc(h, &D.lam_h, Bm));
    std::vector<int> jrow(size_t(e.nnz_jac));
    for (int k = 0; k < e.nnz_jac; ++k) jrow[size_t(k)] = e.jac_i[size_t(k)];
    A_(ipm_alloc_c(h, &D.jac_row, jrow));
    D.nnz_var = e.nnz_nl + e.nnz_lin;
  }
  {   // restoration phase and second-order correction work space (m >= 1 keeps the allocations non-empty)
    const size_t Bm = B * size_t(std::max(p.m, 1));
    A_(ipm_alloc(h, &D.pp, Bm)); A_(ipm_alloc(h, &D.nn, Bm)); A_(ipm_alloc(h, &D.zp, Bm)); A_(ipm_alloc(h, &D.zn, Bm));
    A_(ipm_alloc(h, &D.dpp, Bm)); A_(ipm_alloc(h, &D.dnn, Bm)); A_(ipm_alloc(h, &D.dzp, Bm)); A_(ipm_alloc(h, &D.dzn, Bm));
    A_(ipm_alloc(h, &D.dlam2, Bm)); A_(ipm_alloc(h, &D.csoc, Bm)); A_(ipm_alloc(h, &D.ct, Bm));
    A_(ipm_alloc(h, &D.dv2, B * p.nv)); A_(ipm_alloc(h, &D.dzL2, B * p.nv)); A_(ipm_alloc(h, &D.dzU2, B * p.nv));
    A_(ipm_alloc(h, &D.rfilt, B * 2 * IPM_FMAX));
    A_(ipm_alloc(h, &D.part, B * IPM_VEC_BLOCKS * IPM_VEC_PART)); A_(ipm_alloc(h, &D.tick, B));
    // on the engine's (non-blocking) stream, where the kernels that take tickets run: a null-stream fill is not ordered against it
    if (hipMemsetAsync(D.tick, 0, B * sizeof(int), static_cast<hipStream_t>(dev_stream(h->eng->e))) != hipSuccess) { h->err = "hipMemset"; return fail(RPM_E_DEVICE); }
  }
  {   // factorisation sub-problems: the whole band + border matrix, or the interval blocks followed by the separator system
    std::vector<KktSub> subs;
    if (p.nd) {
      for (const KktSubHost& g : p.subs) subs.push_back(KktSub{KktGeom{g.Nt, g.Nb, g.nb, g.b, g.CS}, g.roff, g.koff});
    } else {
      subs.push_back(KktSub{KktGeom{p.Nt, p.Nb, p.nb, p.b, p.CS}, 0, 0});
    }
    D.n_sub = int(subs.size());
    D.n_l2 = p.nd ? p.n_l2 : 0;
    D.n_l1 = p.nd ? D.n_sub - 1 - D.n_l2 : 0;
    D.max_sub_nt = 0;
    for (const KktSub& q : subs) D.max_sub_nt = std::max(D.max_sub_nt, q.g.Nt);
    D.l1_dense_lds = D.l2_dense_lds = D.last_dense_lds = 0;
    D.last_dense_corner = 1;
    if (D.n_l1 > 0) {
      auto dense_lds_of = [&](int first, int count) -> size_t {   // LDS of kkt_factor_dense_kernel for these sub-problems, 0: one does not fit
        int rows = 0;   // most 16-row blocks (band + border)
        for (int i = first; i < first + count; ++i)
          rows = std::max(rows, (subs[size_t(i)].g.Nb + IPM_W - 1) / IPM_W + (subs[size_t(i)].g.nb + IPM_W - 1) / IPM_W);
        return count > 0 && rows <= kkt_factor_dense_max_block_rows() ? kkt_factor_dense_lds_bytes(rows) : 0;
      };
      h->l1_dense_lds = dense_lds_of(0, D.n_l1);
      h->l2_dense_lds = dense_lds_of(D.n_l1, D.n_l2);
      for (int i = D.n_l1; i < D.n_l1 + D.n_l2; ++i)     // groups of a narrow band stay on kkt_factor_kernel, which skips what lies outside the band
        if (2 * subs[size_t(i)].g.b < subs[size_t(i)].g.Nb) h->l2_dense_lds = 0;
      h->last_dense_lds = dense_lds_of(D.n_l1 + D.n_l2, 1);
      // (one workgroup per CU: a sweep of many small last levels is better off on kkt_factor_kernel, several workgroups per CU —
      // 1024 quadrotor instances 0.12 against 0.30 ms)
      if (B > 256) h->last_dense_lds = 0;
      const size_t most = std::max(h->l1_dense_lds, std::max(h->l2_dense_lds, h->last_dense_lds));
      if (most && kkt_factor_dense_prepare(most) != hipSuccess) { h->err = "hipFuncSetAttribute"; return fail(RPM_E_DEVICE); }
      if (!(std::getenv("RPM_IPM_DENSE") && std::atoi(std::getenv("RPM_IPM_DENSE")) == 0)) D.l1_dense_lds = h->l1_dense_lds;   // option "level1_dense"
      if (!(std::getenv("RPM_IPM_UPPER_DENSE") && std::atoi(std::getenv("RPM_IPM_UPPER_DENSE")) == 0)) {                      // option "upper_dense"
        D.l2_dense_lds = h->l2_dense_lds;
        D.last_dense_lds = h->last_dense_lds;
      }
    }
    // level 1 assembled by kkt_factor_dense_kernel itself (IpmDev::df_on): per interval block the list of its structural slots and,
    // per register tile lane, which of them it holds; the fill leaves out the chunks that lie inside such a block
    D.df_on = 0;
    D.df_ptr = D.df_ki = D.df_hg = D.as_live = nullptr;
    D.as_nlive = 0;
    D.df_map = nullptr;
    D.df_tiles = IPM_DENSE_TILES;
    if (h->l1_dense_lds && D.as_nchunk > 0 && !(std::getenv("RPM_IPM_FUSED_FILL") && std::atoi(std::getenv("RPM_IPM_FUSED_FILL")) == 0)) {
      // per block the Jacobian entries first, then the Hessian slots, then the rest (slack entries, diagonals): three plain loops in the kernel
      std::vector<int> f_ptr(3 * size_t(D.n_l1) + 1, 0), f_ki, f_hg, skip(size_t(D.as_nchunk), 0);
      int df_tiles = IPM_DENSE_TILES;
      for (int si = 0; si < D.n_l1; ++si)
        df_tiles = std::max(df_tiles, ipm_dense_tiles_of((subs[size_t(si)].g.Nb + IPM_W - 1) / IPM_W + (subs[size_t(si)].g.nb + IPM_W - 1) / IPM_W));
      std::vector<unsigned long long> f_map(size_t(D.n_l1) * df_tiles * 64, 0ull);
      bool ok = true;
      size_t e2 = 0;
      for (int si = 0; si < D.n_l1 && ok; ++si) {
        const KktGeom g = subs[size_t(si)].g;
        const long long k0 = subs[size_t(si)].koff, k1 = k0 + (long long)g.Nt * g.CS;
        const int nbb = (g.Nb + IPM_W - 1) / IPM_W, nbr = (g.nb + IPM_W - 1) / IPM_W, NTB = nbb + nbr;
        while (e2 < ents.size() && ents[e2].dst < k0) ++e2;     // (level-1 blocks come first in the storage, in order)
        size_t e3 = e2;
        while (e3 < ents.size() && ents[e3].dst < k1) ++e3;
        int number = 0;
        for (int cls = 0; cls < 3 && ok; ++cls) {
          f_ptr[3 * size_t(si) + size_t(cls)] = int(f_ki.size());
          for (size_t q = e2; q < e3; ++q) {
            const int kind = ents[q].ki >> 28;
            if ((kind == 1 ? 0 : (kind == 0 ? 1 : 2)) != cls) continue;
            const long long o = ents[q].dst - k0;
            const int j = int(o / g.CS), slot = int(o % g.CS);
            const int i = (j < g.Nb && slot <= g.b) ? j + slot : g.Nb + slot - (g.b + 1);
            const bool band = i < g.Nb;
            if (i < j || i >= g.Nt || (band && (j >= g.Nb || i - j > g.b)) || (long long)j * g.CS + (band ? i - j : g.b + 1 + i - g.Nb) != o) { ok = false; break; }
            const int I = band ? i / IPM_W : nbb + (i - g.Nb) / IPM_W, Kb = j < g.Nb ? j / IPM_W : nbb + (j - g.Nb) / IPM_W;
            const int lr = i - (I < nbb ? IPM_W * I : g.Nb + IPM_W * (I - nbb)), cc = j - (Kb < nbb ? IPM_W * Kb : g.Nb + IPM_W * (Kb - nbb));
            const int tile = ipm_dense_tile(NTB, I, Kb);
            if (tile >= df_tiles || ++number > 0xffff) { ok = false; break; }
            f_map[(size_t(si) * df_tiles + tile) * 64 + size_t((cc & 3) * 16 + lr)] |= (unsigned long long)number << (16 * (cc >> 2));
            f_ki.push_back(ents[q].ki);
            f_hg.push_back(ents[q].hg);
          }
        }
        e2 = e3;
        // the values wait in the panel's LDS space (2 x block rows x 16 rows of IPM_DENSE_LDS_ROW doubles), slot 0 is the zero
        ok = ok && size_t(number) + 1 <= 2 * size_t(NTB) * IPM_W * IPM_DENSE_LDS_ROW;
        for (long long c = (k0 + IPM_FILL_CHUNK - 1) / IPM_FILL_CHUNK; ok && (c + 1) * IPM_FILL_CHUNK <= k1; ++c) skip[size_t(c)] = 1;
      }
      f_ptr[3 * size_t(D.n_l1)] = int(f_ki.size());
      if (ok) {
        A_(ipm_alloc_c(h, &D.df_ptr, f_ptr)); A_(ipm_alloc_c(h, &D.df_ki, f_ki)); A_(ipm_alloc_c(h, &D.df_hg, f_hg));
        std::vector<int> live;
        for (int c = 0; c < D.as_nchunk; ++c)
          if (!skip[size_t(c)]) live.push_back(c);
        D.as_nlive = int(live.size());
        A_(ipm_alloc_c(h, &D.as_live, live)); A_(ipm_alloc_c(h, &D.df_map, f_map));
        D.df_tiles = df_tiles;
        D.df_on = 1;
      }
    }
    A_(ipm_alloc_c(h, &D.subs, subs));
    A_(ipm_alloc(h, &D.piv, B * subs.size() * 3));
    A_(ipm_alloc_c(h, &D.cg_ptr, p.cg_ptr)); A_(ipm_alloc_c(h, &D.cg_src, p.cg_src)); A_(ipm_alloc_c(h, &D.cg_dst, p.cg_dst));
    A_(ipm_alloc_c(h, &D.rg_ptr, p.rg_ptr)); A_(ipm_alloc_c(h, &D.rg_src, p.rg_src)); A_(ipm_alloc_c(h, &D.rg_dst, p.rg_dst));
    A_(ipm_alloc_c(h, &D.rs_dst, p.rs_dst)); A_(ipm_alloc_c(h, &D.rs_src, p.rs_src)); A_(ipm_alloc_c(h, &D.gap_pos, p.gap_pos));
    D.n_cg = int(p.cg_dst.size()); D.n_rg = int(p.rg_dst.size()); D.n_rs = int(p.rs_dst.size()); D.n_gap = int(p.gap_pos.size());
    A_(ipm_alloc_c(h, &D.cg2_ptr, p.cg2_ptr)); A_(ipm_alloc_c(h, &D.cg2_src, p.cg2_src)); A_(ipm_alloc_c(h, &D.cg2_dst, p.cg2_dst));
    A_(ipm_alloc_c(h, &D.rg2_ptr, p.rg2_ptr)); A_(ipm_alloc_c(h, &D.rg2_src, p.rg2_src)); A_(ipm_alloc_c(h, &D.rg2_dst, p.rg2_dst));
    A_(ipm_alloc_c(h, &D.rs2_dst, p.rs2_dst)); A_(ipm_alloc_c(h, &D.rs2_src, p.rs2_src));
    D.n_cg2 = int(p.cg2_dst.size()); D.n_rg2 = int(p.rg2_dst.size()); D.n_rs2 = int(p.rs2_dst.size());
    D.n_cg_long = p.n_cg_long; D.n_cg2_long = p.n_cg2_long;
  }
#undef A_
  if (hipHostMalloc(reinterpret_cast<void**>(&h->h_cnt), 4 * sizeof(int)) != hipSuccess) { h->err = "hipHostMalloc"; return fail(RPM_E_DEVICE); }
  // variable bounds of every instance default to the engine's
  {
    std::vector<double> l(B * p.nv, 0.0), u(B * p.nv, 0.0);
    for (size_t bi = 0; bi < B; ++bi)
      for (int i = 0; i < p.n; ++i) { l[bi * p.nv + i] = e.xl[i]; u[bi * p.nv + i] = e.xu[i]; }
    if (hipMemcpy(D.vl0, l.data(), l.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(D.vu0, u.data(), u.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) { h->err = "hipMemcpy"; return fail(RPM_E_DEVICE); }
  }
  h->factor_mt = p.max_rows <= 256 ? 4 : (p.max_rows <= 384 ? 6 : IPM_MT);
  // a few large instances: fewer sub-problems than two per CU -> 8 waves per workgroup, 3 tiles each (code 38: <3, 8>)
  if (p.nd && p.max_rows > 256 && p.max_rows <= 384 && B * p.subs.size() <= 512) h->factor_mt = 38;
  if (const char* fm = std::getenv("RPM_IPM_FACTOR_VARIANT")) h->factor_mt = std::atoi(fm);   // experiments: 4, 6, 8, 28 (<2,8>), 38 (<3,8>)
  if (p.max_rows > 4 * IPM_MT * 16) {
    h->err = "band + border of " + std::to_string(p.max_rows - IPM_W) + " rows exceeds the factorisation's 512 rows per block column";
    return fail(RPM_E_UNSUPPORTED);
  }
  h->factor_lds = kkt_factor_lds_bytes(p);
  if (h->factor_lds > 150 * 1024) {
    h->err = "band of " + std::to_string(p.b) + " and border of " + std::to_string(p.nb) + " rows do not fit the factorisation's LDS";
    return fail(RPM_E_UNSUPPORTED);
  }
  if (kkt_factor_prepare(h->factor_mt, h->factor_lds) != hipSuccess) { h->err = "hipFuncSetAttribute"; return fail(RPM_E_DEVICE); }
  h->h_inst.resize(B);
  for (hipEvent_t& e2 : h->ev)
    if (hipEventCreate(&e2) != hipSuccess) { h->err = "hipEventCreate"; return fail(RPM_E_DEVICE); }
  *out = h;
  return RPM_OK;
}

void rpm_ipm_destroy(rpm_ipm* h) { delete h; }
const char* rpm_ipm_last_error(const rpm_ipm* h) { return h ? h->err.c_str() : "null solver"; }

int rpm_ipm_set_option(rpm_ipm* h, const char* key, double value) {
  if (!h || !key) return RPM_E_INVALID;
  IpmOpts& o = h->D.o;
  const std::string k(key);
  if (k == "tol") o.tol = value;
  else if (k == "max_iter") o.max_iter = int(value);
  else if (k == "mu_init") o.mu_init = value;
  else if (k == "bound_push") o.bound_push = value;
  else if (k == "bound_frac") o.bound_frac = value;
  else if (k == "delta_c") o.delta_c = value;
  else if (k == "max_line_search") o.max_ls = int(value);
  else if (k == "restoration") o.resto = value != 0.0;
  else if (k == "acceptable_tol") o.acceptable_tol = value;
  else if (k == "acceptable_iter") o.acceptable_iter = int(value);
  else if (k == "restoration_max_iter") o.resto_max = int(value);
  else if (k == "bound_relax_factor") { if (!(value >= 0.0)) { h->err = "bound_relax_factor must be >= 0"; return RPM_E_INVALID; } o.bound_relax = value; }
  else if (k == "max_soc") o.max_soc = std::max(0, int(value));
  else if (k == "sigma_cap") o.sigma_cap = value;      // experiment
  else if (k == "init_ls_multipliers") o.init_ls_mult = value != 0.0;
  else if (k == "nlp_scaling") { o.nlp_scaling = value != 0.0; h->D.scal_on = o.nlp_scaling; }
  else if (k == "nlp_scaling_max_gradient") { if (!(value > 0.0)) { h->err = "nlp_scaling_max_gradient must be > 0"; return RPM_E_INVALID; } o.scal_gmax = value; }
  else if (k == "ic_hot_start") o.ic_hot = value != 0.0;
  else if (k == "ic_hot_min") o.ic_hot_min = value;
  else if (k == "mu_strategy") {       // 0 monotone (default), 1 adaptive: LOQO oracle + kkt-error globalisation
    if (value != 0.0 && value != 1.0) { h->err = "mu_strategy: 0 (monotone) or 1 (adaptive)"; return RPM_E_INVALID; }
    o.mu_adaptive = int(value);
  }
  else if (k == "restoration_penalty") o.resto_rho = value;
  else if (k == "level1_dense") {   // level 1 of the nested dissection on kkt_factor_dense_kernel (default where the interval blocks fit it)
    if (value != 0.0 && !h->l1_dense_lds) { h->err = "level1_dense: no nested dissection, or an interval block of more than 21 block rows"; return RPM_E_UNSUPPORTED; }
    h->D.l1_dense_lds = value != 0.0 ? h->l1_dense_lds : 0;
  }
  else if (k == "upper_dense") {    // the levels above the interval blocks on kkt_factor_dense_kernel too (default where their sub-problems fit it)
    h->D.l2_dense_lds = value != 0.0 ? h->l2_dense_lds : 0;
    h->D.last_dense_lds = value != 0.0 ? h->last_dense_lds : 0;
    h->D.last_dense_corner = value == 2.0 ? 0 : 1;   // 2: the last level's corner by kkt_factor_kernel's unblocked elimination
  }
  else if (k == "fused_fill") {     // level-1 blocks assembled inside kkt_factor_dense_kernel (default where that kernel runs and the tables exist)
    if (value != 0.0 && !h->D.df_map) { h->err = "fused_fill: level 1 does not run on kkt_factor_dense_kernel"; return RPM_E_UNSUPPORTED; }
    h->D.df_on = value != 0.0;
  }
  else if (k == "trace") {          // keep the first `value` iterations of every instance (rpm_ipm_get_trace)
    const int cap = int(value);
    if (cap < 0 || cap > 100000) { h->err = "trace: 0 ... 100000 records"; return RPM_E_INVALID; }
    h->D.trace = nullptr;
    h->D.trace_cap = 0;
    if (cap > 0) {
      int rc = ipm_alloc(h, &h->D.trace, size_t(h->D.B) * cap * IPM_TRACE);
      if (rc) return rc;
      h->D.trace_cap = cap;
    }
  }
  else { h->err = "unknown option " + k; return RPM_E_INVALID; }
  return RPM_OK;
}

int rpm_ipm_get_info(rpm_ipm* h, int* kkt_order, int* band_order, int* half_bandwidth, int* border, long long* storage_doubles,
                     int* n_slacks) {
  if (!h) return RPM_E_INVALID;
  if (kkt_order) *kkt_order = h->plan.Nt;
  if (band_order) *band_order = h->plan.Nb;
  if (half_bandwidth) *half_bandwidth = h->plan.b;
  if (border) *border = h->plan.nb;
  if (storage_doubles) *storage_doubles = h->plan.storage();
  if (n_slacks) *n_slacks = h->plan.ns;
  return RPM_OK;
}

/* the factorisation's sub-problems (one without nested dissection; the interval blocks followed by the separator system
 * with it): 5 ints each — order, banded part, border, half bandwidth, doubles per stored column */
int rpm_ipm_get_subproblems(rpm_ipm* h, int capacity, int* geom, int* n_sub) {
  if (!h || !n_sub) return RPM_E_INVALID;
  const IpmPlan& p = h->plan;
  const int n = p.nd ? int(p.subs.size()) : 1;
  *n_sub = n;
  if (!geom) return RPM_OK;
  if (capacity < n) return RPM_E_INVALID;
  for (int i = 0; i < n; ++i) {
    int* g = geom + 5 * i;
    if (p.nd) { g[0] = p.subs[i].Nt; g[1] = p.subs[i].Nb; g[2] = p.subs[i].nb; g[3] = p.subs[i].b; g[4] = p.subs[i].CS; }
    else { g[0] = p.Nt; g[1] = p.Nb; g[2] = p.nb; g[3] = p.b; g[4] = p.CS; }
  }
  return RPM_OK;
}

int rpm_ipm_get_stats(rpm_ipm* h, int* iterations, int* factorizations, int* trial_points) {
  if (!h) return RPM_E_INVALID;
  if (iterations) *iterations = h->total_iterations;
  if (factorizations) *factorizations = h->total_factorizations;
  if (trial_points) *trial_points = h->total_trials;
  return RPM_OK;
}

int rpm_ipm_get_kernel_times(rpm_ipm* h, double* factor_ms, double* substitution_ms) {
  if (!h) return RPM_E_INVALID;
  if (factor_ms) *factor_ms = h->factor_ms;
  if (substitution_ms) *substitution_ms = h->solve_ms;
  return RPM_OK;
}

int rpm_ipm_get_restorations(rpm_ipm* h, int* per_instance) {
  if (!h || !per_instance || h->h_inst.empty()) return RPM_E_INVALID;
  for (int bi = 0; bi < h->D.B; ++bi) per_instance[bi] = h->h_inst[bi].n_resto;
  return RPM_OK;
}

int rpm_ipm_get_trace(rpm_ipm* h, int instance, int capacity, double* records, int* n_records) {
  if (!h || instance < 0 || instance >= h->D.B || !records || !n_records) return RPM_E_INVALID;
  if (!h->D.trace || h->h_inst.empty()) { *n_records = 0; return RPM_OK; }
  const int n = std::min(std::min(h->h_inst[instance].iter, h->D.trace_cap), capacity);
  IPM_TRY(h, hipMemcpy(records, h->D.trace + size_t(instance) * h->D.trace_cap * IPM_TRACE, size_t(n) * IPM_TRACE * sizeof(double),
                       hipMemcpyDeviceToHost));
  *n_records = n;
  return RPM_OK;
}

int rpm_ipm_set_bounds(rpm_ipm* h, int instance, const double* x_l, const double* x_u) {
  if (!h || !x_l || !x_u || instance < 0 || instance >= h->D.B) return RPM_E_INVALID;
  const IpmPlan& p = h->plan;
  for (int i = 0; i < p.n; ++i)
    if ((x_l[i] == x_u[i]) != (p.fixed[i] != 0)) {
      h->err = "rpm_ipm_set_bounds: variable " + std::to_string(i) + " changes between fixed and free (the KKT layout is shared by all instances)";
      return RPM_E_INVALID;
    }
  // caller arrays go through the engine's staging slots (rpm_device.hip), not the runtime's pageable-copy path
  Engine& e = h->eng->e;
  int rc = dev_upload(e, h->D.vl0 + size_t(instance) * p.nv, x_l, size_t(p.n), STAGE_X);
  if (!rc) rc = dev_upload(e, h->D.vu0 + size_t(instance) * p.nv, x_u, size_t(p.n), STAGE_G);
  if (!rc) rc = dev_sync(e);
  if (rc) h->err = e.err;
  return rc;
}

/* variable bounds of all instances at once: x_l, x_u are n_instances x n (host), e.g. the measured initial states of a
 * receding-horizon sweep; two copies instead of 2 n_instances */
int rpm_ipm_set_all_bounds(rpm_ipm* h, const double* x_l, const double* x_u) {
  if (!h || !x_l || !x_u) return RPM_E_INVALID;
  const IpmPlan& p = h->plan;
  const size_t B = size_t(h->D.B);
  for (size_t bi = 0; bi < B; ++bi)
    for (int i = 0; i < p.n; ++i)
      if ((x_l[bi * p.n + i] == x_u[bi * p.n + i]) != (p.fixed[i] != 0)) {
        h->err = "rpm_ipm_set_all_bounds: instance " + std::to_string(bi) + ", variable " + std::to_string(i) +
                 " changes between fixed and free (the KKT layout is shared by all instances)";
        return RPM_E_INVALID;
      }
  Engine& e = h->eng->e;
  hipStream_t st = static_cast<hipStream_t>(dev_stream(e));
  double *sl = nullptr, *su = nullptr;
  int rc = dev_stage_reserve(e, STAGE_X, B * p.n, &sl, nullptr);
  if (!rc) rc = dev_stage_reserve(e, STAGE_G, B * p.n, &su, nullptr);
  if (rc) { h->err = e.err; return rc; }
  std::memcpy(sl, x_l, B * p.n * sizeof(double));
  std::memcpy(su, x_u, B * p.n * sizeof(double));
  IPM_TRY(h, hipMemcpy2DAsync(h->D.vl0, size_t(p.nv) * sizeof(double), sl, size_t(p.n) * sizeof(double), size_t(p.n) * sizeof(double), B,
                              hipMemcpyHostToDevice, st));
  IPM_TRY(h, hipMemcpy2DAsync(h->D.vu0, size_t(p.nv) * sizeof(double), su, size_t(p.n) * sizeof(double), size_t(p.n) * sizeof(double), B,
                              hipMemcpyHostToDevice, st));
  IPM_TRY(h, hipStreamSynchronize(st));
  return RPM_OK;
}

/* test hook: factor + solve the caller's matrices (B x storage doubles in the band + border layout, lower triangle)
 * against B right-hand sides in KKT order; returns the solutions and the signs of D */
int rpm_ipm_debug_solve(rpm_ipm* h, const double* k_storage, const double* rhs, double* sol, int* n_pos, int* n_neg) {
  if (!h || !k_storage || !rhs || !sol) return RPM_E_INVALID;
  const IpmPlan& p = h->plan;
  IpmDev& D = h->D;
  hipStream_t st = static_cast<hipStream_t>(dev_stream(h->eng->e));
  std::vector<IpmInst> inst(D.B);
  for (auto& s : inst) { s = IpmInst{}; s.refactor = 1; }
  IPM_TRY(h, hipMemcpy(D.inst, inst.data(), inst.size() * sizeof(IpmInst), hipMemcpyHostToDevice));
  if (p.nd) { h->err = "rpm_ipm_debug_solve takes the band + border storage; with nested dissection use rpm_ipm_debug_solve_dense"; return RPM_E_UNSUPPORTED; }
  IPM_TRY(h, hipMemcpy(D.K, k_storage, size_t(D.B) * p.storage() * sizeof(double), hipMemcpyHostToDevice));
  IPM_TRY(h, hipMemcpy(D.rhs, rhs, size_t(D.B) * p.Nt * sizeof(double), hipMemcpyHostToDevice));
  int rc = factor_and_solve_launch(h, st, true, true, 0);
  if (rc) return rc;
  ipm_launch_inertia(D, st);      // sums the pivot signs into the instance records
  IPM_TRY(h, hipStreamSynchronize(st));
  IPM_TRY(h, hipMemcpy(sol, D.rhs, size_t(D.B) * p.Nt * sizeof(double), hipMemcpyDeviceToHost));
  IPM_TRY(h, hipMemcpy(inst.data(), D.inst, inst.size() * sizeof(IpmInst), hipMemcpyDeviceToHost));
#ifdef IPM_TIMING
  fprintf(stderr, "factor phases of instance 0 [100 MHz ticks]: T %lld  k-loop %lld  diag %lld  panel %lld  corner %lld  tail %lld\n",
          inst[0].dbg[0], inst[0].dbg[1], inst[0].dbg[2], inst[0].dbg[3], inst[0].dbg[4], inst[0].dbg[5]);
#endif
  for (int bi = 0; bi < D.B; ++bi) {
    if (n_pos) n_pos[bi] = inst[bi].npos;
    if (n_neg) n_neg[bi] = inst[bi].nneg;
  }
  return RPM_OK;
}

/* test hook, layout-independent: factor + solve the caller's DENSE symmetric matrices (B x Nt x Nt, row-major, rows and columns
 * in unknown order: [0,n) variables, slacks, multipliers) against B right-hand sides (unknown order); entries the layout has
 * no slot for must be zero (RPM_E_INVALID otherwise).  Works for the band + border layout and for nested dissection. */
int rpm_ipm_debug_solve_dense(rpm_ipm* h, const double* k_dense, const double* rhs, double* sol, int* n_pos, int* n_neg) {
  if (!h || !k_dense || !rhs || !sol) return RPM_E_INVALID;
  const IpmPlan& p = h->plan;
  IpmDev& D = h->D;
  hipStream_t st = static_cast<hipStream_t>(dev_stream(h->eng->e));
  const size_t B = size_t(D.B), Nt = size_t(p.Nt);
  std::vector<double> store(B * size_t(p.storage()), 0.0), r(B * size_t(p.Nt_alloc), 0.0);
  for (size_t bi = 0; bi < B; ++bi)
    for (size_t a = 0; a < Nt; ++a) {
      r[bi * p.Nt_alloc + p.pos[a]] = rhs[bi * Nt + a];
      for (size_t c = 0; c <= a; ++c) {
        const double v = k_dense[(bi * Nt + a) * Nt + c];
        if (v == 0.0) continue;
        const long long o = ipm_plan_offset(p, int(a), int(c));
        if (o < 0) { h->err = "rpm_ipm_debug_solve_dense: entry (" + std::to_string(a) + ", " + std::to_string(c) + ") has no slot in the layout"; return RPM_E_INVALID; }
        store[bi * size_t(p.storage()) + size_t(o)] = v;
      }
    }
  std::vector<IpmInst> inst(D.B);
  for (auto& s2 : inst) { s2 = IpmInst{}; s2.refactor = 1; }
  IPM_TRY(h, hipMemcpy(D.inst, inst.data(), inst.size() * sizeof(IpmInst), hipMemcpyHostToDevice));
  IPM_TRY(h, hipMemcpy(D.K, store.data(), store.size() * sizeof(double), hipMemcpyHostToDevice));
  IPM_TRY(h, hipMemcpy(D.rhs, r.data(), r.size() * sizeof(double), hipMemcpyHostToDevice));
  const int df_keep = D.df_on;
  D.df_on = 0;                 // factor what is in the storage, not the solver's own matrix
  int rc = factor_and_solve_launch(h, st, true, true, 0);
  D.df_on = df_keep;
  if (rc) return rc;
  ipm_launch_inertia(D, st);
  IPM_TRY(h, hipStreamSynchronize(st));
  IPM_TRY(h, hipMemcpy(r.data(), D.rhs, r.size() * sizeof(double), hipMemcpyDeviceToHost));
  IPM_TRY(h, hipMemcpy(inst.data(), D.inst, inst.size() * sizeof(IpmInst), hipMemcpyDeviceToHost));
#ifdef IPM_TIMING
  // kkt_factor_dense_kernel (build with -DIPM_TIMING_SUB=<out of range>): tile wave 0 and the diagonal wave of interval block 0
  fprintf(stderr, "level-1 phases of instance 0 [100 MHz ticks]: panel %lld  wait B3 %lld  next diagonal tile + B1 %lld  update %lld  wait B2 %lld  early block columns %lld | diagonal wave: waiting %lld  factoring %lld\n",
          inst[0].dbg[0], inst[0].dbg[1], inst[0].dbg[2], inst[0].dbg[3], inst[0].dbg[4], inst[0].dbg[5], inst[0].dbg[6], inst[0].dbg[7]);
  // kkt_factor_kernel (-DIPM_TIMING_SUB=<sub-problem>): the same record read as the left-looking kernel's phases
  fprintf(stderr, "left-looking phases of instance 0 [100 MHz ticks]: T %lld  k-loop %lld  diag %lld  panel %lld  corner %lld  tail %lld\n",
          inst[0].dbg[0], inst[0].dbg[1], inst[0].dbg[2], inst[0].dbg[3], inst[0].dbg[4], inst[0].dbg[5]);
#endif
  for (size_t bi = 0; bi < B; ++bi) {
    for (size_t a = 0; a < Nt; ++a) sol[bi * Nt + a] = r[bi * p.Nt_alloc + p.pos[a]];
    if (n_pos) n_pos[bi] = inst[bi].npos;
    if (n_neg) n_neg[bi] = inst[bi].nneg;
  }
  return RPM_OK;
}

/* storage offset of the entry between unknowns ua and uc (unknown order as above), -1 if the layout has no slot for it */
int rpm_ipm_debug_slot(rpm_ipm* h, int ua, int uc, long long* offset) {
  if (!h || !offset || ua < 0 || uc < 0 || ua >= h->plan.Nt || uc >= h->plan.Nt) return RPM_E_INVALID;
  *offset = ipm_plan_offset(h->plan, ua, uc);
  return RPM_OK;
}

/* KKT position of every unknown ([0,n) variables, then the slacks, then the m multipliers) — for tests and tools */
int rpm_ipm_get_permutation(rpm_ipm* h, int* pos, int capacity) {
  if (!h || !pos || capacity < h->plan.Nt) return RPM_E_INVALID;
  std::memcpy(pos, h->plan.pos.data(), sizeof(int) * h->plan.Nt);
  return RPM_OK;
}

int rpm_ipm_solve_dev(rpm_ipm* h, double* d_x, double* d_lambda, double* obj, int* status, int* iterations, double* kkt_error,
                      void* stream) {
  if (!h || !d_x) return RPM_E_INVALID;
  Engine& e = h->eng->e;
  IpmDev& D = h->D;
  const IpmPlan& p = h->plan;
  hipStream_t st = static_cast<hipStream_t>(dev_stream(e));
  const unsigned B = unsigned(D.B);
  auto eng_fail = [&](int rc) { h->err = e.err; return rc; };
  dev_forget_persistent(e);   // the first Jacobian evaluation of this solve writes the constant block of D.jac, the later ones skip it
  if (h->lbfgs) lb_launch_reset(D, st);
  h->lb_iterations = 0;
  h->total_factorizations = h->total_iterations = h->total_trials = h->total_soc = 0;
  h->factor_ms = h->solve_ms = 0.0;
  h->solve_pending = false;
  if (D.sg != e.stride_g() || D.sv != e.stride_values()) {   // rpm_set_option refuses this while a solver is attached; belt and braces
    h->err = "rpm_ipm_solve_dev: the engine's instance strides changed after rpm_ipm_create";
    return RPM_E_INVALID;
  }
  // Ordering contract (rpm_hip.h): the loop runs on the engine's private stream.  Its first read of d_x / its first write of
  // d_lambda wait for everything the caller queued on `stream` before this call; the call returns after the solver's stream
  // has drained, so the results are complete for every stream and for the host.
  IPM_TRY(h, hipEventRecord(h->ev[0], static_cast<hipStream_t>(stream)));
  IPM_TRY(h, hipStreamWaitEvent(st, h->ev[0], 0));

  IPM_TRY(h, hipMemcpyAsync(D.xt, d_x, size_t(B) * p.n * sizeof(double), hipMemcpyDeviceToDevice, st));
  ipm_launch_init(D, d_x, st);
  int rc;
  const bool scal = D.scal_on != 0;
  if (scal) {   // Ipopt's gradient-based scaling: factors from the gradients at the caller's starting point (before it is pushed inside its bounds)
    if ((rc = dev_eval_obj(e, D.xt, D.objt, D.grad, st))) return eng_fail(rc);
    if ((rc = dev_eval_cons(e, D.xt, D.gt, D.jac, 3 | 4 | 16, st))) return eng_fail(rc);
    ipm_launch_scaling_factors(D, st);
    // this evaluation has written D.jac's constant block, which the later ones leave alone: it is scaled here, once
    if (e.nnz_jac > D.nnz_var) ipm_launch_scale(D, nullptr, D.jac, D.nnz_var, e.nnz_jac, nullptr, nullptr, st);
  }
  ipm_launch_pack_x(D, st);
  rc = dev_eval_cons(e, D.xe, D.g, nullptr, 1 | 4, st);
  if (rc) return eng_fail(rc);
  if (scal) ipm_launch_scale(D, D.g, nullptr, 0, 0, nullptr, nullptr, st);
  ipm_launch_init_slack(D, st);
  if ((rc = launch_check(h, "ipm_init"))) return rc;

  for (;;) {
    ipm_launch_pack_x(D, st);
    if ((rc = dev_eval_obj(e, D.xe, D.obj, D.grad, st))) return eng_fail(rc);
    // D.jac is this solver's own array, written by the tile kernel only: its constant Doffdiag block (55 % of the metric
    // problem's Jacobian) is written by the first evaluation of a solve and left alone afterwards (flag 16)
    if ((rc = dev_eval_cons(e, D.xe, D.g, D.jac, 3 | 4 | 16, st))) return eng_fail(rc);
    // (without the scaling's own evaluation the first pass of a solve writes the constant block too: then all of D.jac is scaled)
    if (scal) ipm_launch_scale(D, D.g, D.jac, 0, D.nnz_var, D.obj, D.grad, st);
    IPM_TRY(h, hipMemsetAsync(D.cnt, 0, 4 * sizeof(int), st));
    ipm_launch_residual(D, st);
    if (h->lbfgs) lb_launch_update(D, st);     // the pair of the step just taken (grad_x L at the new point is in D.glag)
    if ((rc = fetch_counts(h, st))) return rc;
    if (h->h_cnt[0] == 0) break;
    h->total_iterations += 1;
    if (!h->lbfgs) {
      if (scal) ipm_launch_scale_lambda(D, st);            // sf (H_f + sum (lambda_i sc_i / sf) H_ci)
      if ((rc = dev_eval_h(e, D.xe, 1.0, scal ? D.lam_h : D.lam, D.hess, st))) return eng_fail(rc);
      if (scal) ipm_launch_scale_hessian(D, st);
    }
    for (int tries = 0; tries < 80; ++tries) {
      IPM_TRY(h, hipMemsetAsync(D.cnt + 1, 0, sizeof(int), st));
      ipm_launch_assemble(D, std::max(e.nnz_jac, e.nnz_h), st);
      IPM_TRY(h, hipEventRecord(h->ev[0], st));
      if ((rc = factor_and_solve_launch(h, st, true, false, 1))) return rc;
      IPM_TRY(h, hipEventRecord(h->ev[1], st));
      ipm_launch_inertia(D, st);
      h->total_factorizations += 1;
      if ((rc = fetch_counts(h, st))) return rc;
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, h->ev[0], h->ev[1]) == hipSuccess) h->factor_ms += ms;
      if (h->h_cnt[1] == 0) break;
    }
    IPM_TRY(h, hipEventRecord(h->ev[2], st));
    if (h->lbfgs && h->lb_iterations > 0) {
      // Z = K0^-1 E with the factors in place (all columns of all instances in one pass), then C = M - E'Z
      lb_launch_columns_and_solve(D, st);
      lb_launch_small(D, st);
    }
    if ((rc = factor_and_solve_launch(h, st, false, true, 1, 1))) return rc;   // (the right-hand side the factorisation was given)
    if (h->lbfgs) lb_launch_correct(D, 1, st);
    IPM_TRY(h, hipEventRecord(h->ev[3], st));
    h->solve_pending = true;
    IPM_TRY(h, hipMemsetAsync(D.cnt + 2, 0, 2 * sizeof(int), st));
    ipm_launch_direction(D, st);
    // line search rounds: every pending instance evaluates one trial point per round — its next backtracking step, or the
    // next second-order correction (right-hand side, substitution with the factors in place, step lengths) where one was asked for
    h->h_cnt[3] = 0;
    for (int ls = 0; ls <= D.o.max_ls + D.o.max_soc + 2; ++ls) {
      if (h->h_cnt[3] > 0) {
        ipm_launch_soc_rhs(D, st);
        if ((rc = factor_and_solve_launch(h, st, false, true, 2))) return rc;
        if (h->lbfgs) lb_launch_correct(D, 2, st);
        ipm_launch_soc_direction(D, st);
        h->total_soc += 1;
      }
      ipm_launch_trial(D, st);
      if ((rc = dev_eval_obj(e, D.xt, D.objt, nullptr, st))) return eng_fail(rc);
      if ((rc = dev_eval_cons(e, D.xt, D.gt, nullptr, 1 | 4, st))) return eng_fail(rc);
      if (scal) ipm_launch_scale(D, D.gt, nullptr, 0, 0, D.objt, nullptr, st);
      IPM_TRY(h, hipMemsetAsync(D.cnt + 2, 0, 2 * sizeof(int), st));
      ipm_launch_accept(D, st);
      h->total_trials += 1;
      if ((rc = fetch_counts(h, st))) return rc;
      if (h->solve_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, h->ev[2], h->ev[3]) == hipSuccess) h->solve_ms += ms;
        h->solve_pending = false;
      }
      if (h->h_cnt[2] == 0 && h->h_cnt[3] == 0) break;
    }
    ipm_launch_update(D, st);
    if (h->lbfgs) {
      ipm_launch_jt_lambda_into(D, D.lb_gold, st);   // grad_x L(x_old, lambda_new): D.grad / D.jac still belong to the old iterate
      h->lb_iterations += 1;
    }
    if ((rc = launch_check(h, "ipm iteration"))) return rc;
  }
  // results: x back into the caller's array, multipliers, per-instance verdicts
  ipm_launch_pack_x(D, st);
  IPM_TRY(h, hipMemcpyAsync(d_x, D.xe, size_t(B) * p.n * sizeof(double), hipMemcpyDeviceToDevice, st));
  std::vector<double> sf_host;
  if (d_lambda) {
    if (scal) ipm_launch_unscale_lambda(D, d_lambda, st);     // the multipliers of the caller's (unscaled) rows
    else IPM_TRY(h, hipMemcpyAsync(d_lambda, D.lam, size_t(B) * p.m * sizeof(double), hipMemcpyDeviceToDevice, st));
  }
  if (scal) {
    sf_host.resize(B);
    IPM_TRY(h, hipMemcpyAsync(sf_host.data(), D.sf, size_t(B) * sizeof(double), hipMemcpyDeviceToHost, st));
  }
  IPM_TRY(h, hipMemcpyAsync(h->h_inst.data(), D.inst, size_t(B) * sizeof(IpmInst), hipMemcpyDeviceToHost, st));
  IPM_TRY(h, hipStreamSynchronize(st));
#ifdef IPM_TIMING
  fprintf(stderr, "last factorisation of instance 0, phase clocks [100 MHz ticks]: T %lld  k-loop %lld  diag %lld  panel %lld  corner %lld  tail %lld | dense level 1, diagonal wave: waiting %lld  factoring %lld\n",
          h->h_inst[0].dbg[0], h->h_inst[0].dbg[1], h->h_inst[0].dbg[2], h->h_inst[0].dbg[3], h->h_inst[0].dbg[4], h->h_inst[0].dbg[5], h->h_inst[0].dbg[6], h->h_inst[0].dbg[7]);
#endif
  for (unsigned bi = 0; bi < B; ++bi) {
    const IpmInst& S = h->h_inst[bi];
    if (obj) obj[bi] = scal ? S.f / sf_host[bi] : S.f;
    if (status) status[bi] = S.status == 1 ? 0 : (S.status == 6 ? 1 : S.status);
    if (iterations) iterations[bi] = S.iter;
    if (kkt_error) kkt_error[bi] = S.err0;
  }
  return RPM_OK;
}

int rpm_ipm_solve(rpm_ipm* h, double* x, double* lambda, double* obj, int* status, int* iterations, double* kkt_error) {
  if (!h || !x) return RPM_E_INVALID;
  IpmDev& D = h->D;
  const IpmPlan& p = h->plan;
  double *d_x = nullptr, *d_l = nullptr;
  IPM_TRY(h, hipMalloc(reinterpret_cast<void**>(&d_x), size_t(D.B) * p.n * sizeof(double)));
  if (hipMalloc(reinterpret_cast<void**>(&d_l), size_t(D.B) * std::max(p.m, 1) * sizeof(double)) != hipSuccess) {
    (void)hipFree(d_x);
    h->err = "hipMalloc";
    return RPM_E_DEVICE;
  }
  // x / lambda are the caller's arrays: through the engine's staging slots (or its page-lock registrations)
  Engine& e = h->eng->e;
  int rc = dev_upload(e, d_x, x, size_t(D.B) * p.n, STAGE_X);
  if (!rc) rc = dev_sync(e);                                                                   // nothing in flight when the solve starts
  if (rc) h->err = e.err;
  if (!rc) rc = rpm_ipm_solve_dev(h, d_x, d_l, obj, status, iterations, kkt_error, nullptr);
  if (!rc) {
    rc = dev_download(e, x, d_x, size_t(D.B) * p.n, STAGE_X);
    if (!rc && lambda) rc = dev_download(e, lambda, d_l, size_t(D.B) * p.m, STAGE_LAMBDA);
    if (rc) h->err = e.err;
  }
  (void)hipFree(d_x);
  (void)hipFree(d_l);
  return rc;
}

}  // extern "C"

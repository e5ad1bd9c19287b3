// rpm_abi.cpp — the extern "C" boundary declared in include/rpm_hip.h.  Thin: argument checks,
// the TNLP value/structure protocol (Core/LpopcIpopt.cpp:11-246), host<->device staging for
// the host-pointer path, and error capture (no exception leaves this file).
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
#include <new>
#include <string>

#include "rpm_engine.hpp"
#include "rpm_pin.h"

static thread_local std::string g_create_error;

using rpm::Engine;

#define RPM_GUARD_BEGIN try {
#define RPM_GUARD_END(eng)                                   \
  }                                                          \
  catch (const std::exception& ex) {                         \
    (eng).err = std::string("internal error: ") + ex.what(); \
    return RPM_E_INVALID;                                    \
  }                                                          \
  catch (...) {                                              \
    (eng).err = "internal error";                            \
    return RPM_E_INVALID;                                    \
  }

static int fail(Engine& e, int code, const char* msg) {
  e.err = msg;
  return code;
}

extern "C" {

int rpm_create(const rpm_problem_desc* desc, rpm_engine** out) {
  if (!out) {
    g_create_error = "rpm_create: out is NULL";
    return RPM_E_INVALID;
  }
  *out = nullptr;
  rpm_engine* h = new (std::nothrow) rpm_engine();
  if (!h) {
    g_create_error = "out of memory";
    return RPM_E_INVALID;
  }
  int rc;
  try {
    rc = rpm::setup_engine(h->e, desc);
  } catch (const std::exception& ex) {
    h->e.err = std::string("internal error: ") + ex.what();
    rc = RPM_E_INVALID;
  }
  if (rc != RPM_OK) {
    g_create_error = h->e.err;
    delete h;
    return rc;
  }
  *out = h;
  return RPM_OK;
}

void rpm_destroy(rpm_engine* h) {
  if (!h) return;
  rpm::device_destroy(h->e);
  delete h;
}

// the reason of the last failed call; when no call failed, the reason the page-lock registry last refused this engine
// (the call itself went through the staging buffers and succeeded)
const char* rpm_last_error(const rpm_engine* h) {
  if (!h) return g_create_error.c_str();
  if (h->e.pin_note.empty()) return h->e.err.c_str();
  rpm_engine* w = const_cast<rpm_engine*>(h);
  w->e.err_report = h->e.err + (h->e.err.empty() ? "" : "; ") + "page-lock registry: " + h->e.pin_note;
  return w->e.err_report.c_str();
}

int rpm_device_init(rpm_engine* h, int device_id) {
  if (!h) return RPM_E_INVALID;
  RPM_GUARD_BEGIN
  return rpm::device_init(h->e, device_id);
  RPM_GUARD_END(h->e)
}

int rpm_get_nlp_info(rpm_engine* h, int* n, int* m, int* nnz_jac_g, int* nnz_h_lag, int* index_style) {
  if (!h) return RPM_E_INVALID;
  Engine& e = h->e;
  RPM_GUARD_BEGIN
  if (e.hessian_mode == RPM_HESSIAN_EXACT && nnz_h_lag) {
    // the Hessian pattern comes from a NaN-propagation probe of the device functor (LpDerivDependciesChecker.cpp),
    // so in exact mode the structure needs the GPU once per mesh
    int rc = rpm::ensure_hessian(e);
    if (rc) return rc;
  }
  if (n) *n = e.n;
  if (m) *m = e.m;
  if (nnz_jac_g) *nnz_jac_g = e.nnz_jac;
  if (nnz_h_lag) *nnz_h_lag = e.nnz_h;
  if (index_style) *index_style = 0;  // TNLP::C_STYLE, LpopcIpopt.cpp:22
  return RPM_OK;
  RPM_GUARD_END(e)
}

int rpm_get_bounds_info(rpm_engine* h, int n, double* x_l, double* x_u, int m, double* g_l, double* g_u) {
  if (!h) return RPM_E_INVALID;
  Engine& e = h->e;
  if (n != e.n || m != e.m || !x_l || !x_u || !g_l || !g_u) return fail(e, RPM_E_INVALID, "get_bounds_info: size mismatch");
  std::memcpy(x_l, e.xl.data(), sizeof(double) * e.n);
  std::memcpy(x_u, e.xu.data(), sizeof(double) * e.n);
  std::memcpy(g_l, e.gl.data(), sizeof(double) * e.m);
  std::memcpy(g_u, e.gu.data(), sizeof(double) * e.m);
  return RPM_OK;
}

int rpm_get_starting_point(rpm_engine* h, int n, int init_x, double* x, int init_z, double* z_L, double* z_U,
                           int m, int init_lambda, double* lambda) {
  (void)z_L; (void)z_U; (void)lambda;
  if (!h) return RPM_E_INVALID;
  Engine& e = h->e;
  // the reference asserts exactly this combination (LpopcIpopt.cpp:86-88)
  if (!init_x || init_z || init_lambda) return fail(e, RPM_E_INVALID, "get_starting_point: only init_x is supported");
  if (n != e.n || m != e.m || !x) return fail(e, RPM_E_INVALID, "get_starting_point: size mismatch");
  std::memcpy(x, e.guess.data(), sizeof(double) * e.n);
  return RPM_OK;
}

// ---- host-pointer evaluations (Ipopt owns every buffer; x/g/values cross PCIe each call): rpm_host_path.hip ----
static int stage_x(Engine& e, int n, const double* x) {   // x into the engine's HBM copy (eval_h)
  if (n != e.n || !x) return fail(e, RPM_E_INVALID, "x size mismatch");
  if (!e.dev) {
    int rc = rpm::device_init(e, 0);
    if (rc) return rc;
  }
  return rpm::dev_upload_x(e, x);
}

int rpm_eval_f(rpm_engine* h, int n, const double* x, int new_x, double* obj_value) {
  if (!h) return RPM_E_INVALID;
  Engine& e = h->e;
  RPM_GUARD_BEGIN
  if (!obj_value) return fail(e, RPM_E_INVALID, "eval_f: obj_value is NULL");
  if (n != e.n || !x) return fail(e, RPM_E_INVALID, "x size mismatch");
  return rpm::host_eval_f(e, x, new_x, obj_value);
  RPM_GUARD_END(e)
}

int rpm_eval_grad_f(rpm_engine* h, int n, const double* x, int new_x, double* grad_f) {
  if (!h) return RPM_E_INVALID;
  Engine& e = h->e;
  RPM_GUARD_BEGIN
  if (!grad_f) return fail(e, RPM_E_INVALID, "eval_grad_f: grad_f is NULL");
  if (n != e.n || !x) return fail(e, RPM_E_INVALID, "x size mismatch");
  return rpm::host_eval_grad_f(e, x, new_x, grad_f);
  RPM_GUARD_END(e)
}

int rpm_eval_g(rpm_engine* h, int n, const double* x, int new_x, int m, double* g) {
  if (!h) return RPM_E_INVALID;
  Engine& e = h->e;
  RPM_GUARD_BEGIN
  if (n != e.n || !x) return fail(e, RPM_E_INVALID, "x size mismatch");
  if (m != e.m || !g) return fail(e, RPM_E_INVALID, "eval_g: size mismatch");
  // x is re-read on every eval_g: Ipopt may call eval_g(new_x=false) after eval_f(new_x=true) on the same x, but
  // the read is cheap next to the results' trip back and keeps the cache logic simple.
  // (new_x only tells the objective's cache whether it is still valid)
  return rpm::host_eval_g(e, x, new_x, g);
  RPM_GUARD_END(e)
}

int rpm_eval_jac_g(rpm_engine* h, int n, const double* x, int new_x, int m, int nele_jac, int* iRow, int* jCol,
                   double* values) {
  if (!h) return RPM_E_INVALID;
  Engine& e = h->e;
  RPM_GUARD_BEGIN
  if (n != e.n || m != e.m || nele_jac != e.nnz_jac) return fail(e, RPM_E_INVALID, "eval_jac_g: size mismatch");
  if (!values) {  // structure pass, LpopcIpopt.cpp:156-164
    if (!iRow || !jCol) return fail(e, RPM_E_INVALID, "eval_jac_g: iRow/jCol are NULL in the structure pass");
    std::memcpy(iRow, e.jac_i.data(), sizeof(int) * e.nnz_jac);
    std::memcpy(jCol, e.jac_j.data(), sizeof(int) * e.nnz_jac);
    return RPM_OK;
  }
  if (!x) return fail(e, RPM_E_INVALID, "eval_jac_g: x is NULL");
  return rpm::host_eval_jac_values(e, x, new_x, values);
  RPM_GUARD_END(e)
}

int rpm_eval_pair(rpm_engine* h, int n, const double* x, int m, double* g, int nele_jac, double* values) {
  if (!h) return RPM_E_INVALID;
  Engine& e = h->e;
  RPM_GUARD_BEGIN
  if (n != e.n || m != e.m || nele_jac != e.nnz_jac || !x || !g || !values) return fail(e, RPM_E_INVALID, "eval_pair: size mismatch or NULL pointer");
  return rpm::host_eval_pair(e, x, g, values);
  RPM_GUARD_END(e)
}

int rpm_eval_h(rpm_engine* h, int n, const double* x, int new_x, double obj_factor, int m, const double* lambda,
               int new_lambda, int nele_hess, int* iRow, int* jCol, double* values) {
  (void)new_lambda;
  if (!h) return RPM_E_INVALID;
  Engine& e = h->e;
  RPM_GUARD_BEGIN
  // With hessian-approximation=limited-memory (the reference's default, LpNLPWrapper.hpp:71) Ipopt never calls eval_h.
  if (e.hessian_mode != RPM_HESSIAN_EXACT) return fail(e, RPM_E_UNSUPPORTED, "eval_h: the engine was created with hessian-approximation=limited-memory");
  int rc = rpm::ensure_hessian(e);
  if (rc) return rc;
  if (n != e.n || m != e.m || nele_hess != e.nnz_h) return fail(e, RPM_E_INVALID, "eval_h: size mismatch");
  if (!values) {  // structure pass, LpopcIpopt.cpp:187-195
    if (!iRow || !jCol) return fail(e, RPM_E_INVALID, "eval_h: iRow/jCol are NULL in the structure pass");
    std::memcpy(iRow, e.hes_i.data(), sizeof(int) * e.nnz_h);
    std::memcpy(jCol, e.hes_j.data(), sizeof(int) * e.nnz_h);
    return RPM_OK;
  }
  if (!x || !lambda) return fail(e, RPM_E_INVALID, "eval_h: x or lambda is NULL");
  if (new_x) rpm::host_new_x(e);
  rc = stage_x(e, n, x);
  if (rc) return rc;
  // the reference copies only m-1 multipliers (LpopcIpopt.cpp:205-208); the last one belongs to a linear row and
  // never enters the Hessian, so all m are uploaded here
  rc = rpm::dev_upload(e, rpm::dev_buf(e, 5), lambda, size_t(e.n_instances) * e.m, rpm::STAGE_LAMBDA);
  if (rc) return rc;
  rc = rpm::dev_eval_h(e, rpm::dev_buf(e, 0), obj_factor, rpm::dev_buf(e, 5), rpm::dev_buf(e, 6), rpm::dev_stream(e));
  if (rc) return rc;
  rc = rpm::dev_download(e, values, rpm::dev_buf(e, 6), size_t(e.n_instances) * e.nnz_h, rpm::STAGE_HESS);
  if (rc) return rc;
  if (e.opt_check_finite && rpm::dev_nonfinite(e, rpm::dev_buf(e, 6), size_t(e.n_instances) * e.nnz_h) != 0) return fail(e, RPM_E_NONFINITE, "eval_h: non-finite Hessian value");
  return RPM_OK;
  RPM_GUARD_END(e)
}

int rpm_finalize_solution(rpm_engine* h, int status, int n, const double* x, const double* z_L, const double* z_U,
                          int m, const double* g, const double* lambda, double obj_value) {
  (void)status; (void)z_L; (void)z_U; (void)g;
  if (!h) return RPM_E_INVALID;
  Engine& e = h->e;
  if (n != e.n || m != e.m || !x || !lambda) return fail(e, RPM_E_INVALID, "finalize_solution: size mismatch");
  e.sol_x.assign(x, x + n);            // Data_->nlpreturn_x, LpopcIpopt.cpp:237-238
  e.sol_lambda.assign(lambda, lambda + m);
  e.sol_obj = obj_value;
  e.has_solution = true;
  return RPM_OK;
}

int rpm_get_solution(rpm_engine* h, int n, double* x, int m, double* lambda, double* obj_value) {
  if (!h) return RPM_E_INVALID;
  Engine& e = h->e;
  if (!e.has_solution) return fail(e, RPM_E_INVALID, "no solution stored");
  if (n != e.n || m != e.m) return fail(e, RPM_E_INVALID, "get_solution: size mismatch");
  if (x) std::memcpy(x, e.sol_x.data(), sizeof(double) * n);
  if (lambda) std::memcpy(lambda, e.sol_lambda.data(), sizeof(double) * m);
  if (obj_value) *obj_value = e.sol_obj;
  return RPM_OK;
}

// ---- solution extraction --------------------------------------------------
int rpm_nlp2op_control(rpm_engine* h, int phase, const double* x, const double* lambda, double* time, double* state,
                       double* control, double* costate, double* pathmult, double* hamiltonian, double* mayer_cost,
                       double* lagrange_cost) {
  if (!h) return RPM_E_INVALID;
  Engine& e = h->e;
  RPM_GUARD_BEGIN
  if (phase < 0 || phase >= e.P) return fail(e, RPM_E_INVALID, "The phase index is out of rang");
  if (e.n_instances != 1) return fail(e, RPM_E_UNSUPPORTED, "nlp2op_control: one instance per engine");
  if (!x || !lambda) {   // Data_->nlpreturn_x / nlpreturn_lambda, stored by finalize_solution
    if (!e.has_solution) return fail(e, RPM_E_INVALID, "nlp2op_control: no x/lambda given and no solution stored");
    x = e.sol_x.data();
    lambda = e.sol_lambda.data();
  }
  return rpm::dev_nlp2op(e, phase, x, lambda, time, state, control, costate, pathmult, hamiltonian, mayer_cost, lagrange_cost);
  RPM_GUARD_END(e)
}

// ---- mesh-error estimate and ph refinement ----------------------------------
int rpm_solution_error(rpm_engine* h, int phase, const double* x, double* rel_err, int* rows) {
  if (!h) return RPM_E_INVALID;
  Engine& e = h->e;
  RPM_GUARD_BEGIN
  if (phase < 0 || phase >= e.P) return fail(e, RPM_E_INVALID, "The phase index is out of rang");
  if (e.n_instances != 1) return fail(e, RPM_E_UNSUPPORTED, "solution_error: one instance per engine");
  if (!x) {
    if (!e.has_solution) return fail(e, RPM_E_INVALID, "solution_error: no x given and no solution stored");
    x = e.sol_x.data();
  }
  if (rows) *rows = e.ph[phase].N + e.ph[phase].K + 1;
  if (!rel_err) return RPM_OK;
  return rpm::dev_solution_error(e, phase, x, rel_err);
  RPM_GUARD_END(e)
}

static int refine_from(Engine& e, int phase, const double* rel, double tol, int nmin, int nmax, int capacity,
                       double* new_mesh_points, int* new_nodes_per_interval, int* new_n_intervals, double* interval_error,
                       int* no_more_refine) {
  std::vector<double> mesh, emax;
  std::vector<int> nodes;
  const bool done = rpm::ph_refine(e.ph[phase], rel, tol, nmin, nmax, mesh, nodes, emax);
  if (new_n_intervals) *new_n_intervals = int(nodes.size());
  if (no_more_refine) *no_more_refine = done ? 1 : 0;
  if (interval_error) std::memcpy(interval_error, emax.data(), sizeof(double) * emax.size());
  if (new_mesh_points || new_nodes_per_interval) {
    if (capacity < int(nodes.size())) return fail(e, RPM_E_INVALID, "ph_refine_mesh: capacity is smaller than the new interval count");
    if (new_mesh_points) std::memcpy(new_mesh_points, mesh.data(), sizeof(double) * mesh.size());
    if (new_nodes_per_interval) std::memcpy(new_nodes_per_interval, nodes.data(), sizeof(int) * nodes.size());
  }
  return RPM_OK;
}

int rpm_ph_refine_from_error(rpm_engine* h, int phase, const double* rel_err, double tol, int nmin, int nmax, int capacity,
                             double* new_mesh_points, int* new_nodes_per_interval, int* new_n_intervals,
                             double* interval_error, int* no_more_refine) {
  if (!h) return RPM_E_INVALID;
  Engine& e = h->e;
  RPM_GUARD_BEGIN
  if (phase < 0 || phase >= e.P) return fail(e, RPM_E_INVALID, "The phase index is out of rang");
  if (!rel_err) return fail(e, RPM_E_INVALID, "ph_refine_from_error: rel_err is NULL");
  if (!(tol > 0) || nmin < 2 || nmax < nmin) return fail(e, RPM_E_INVALID, "ph_refine_mesh: need tol > 0 and 2 <= Nmin <= Nmax");
  return refine_from(e, phase, rel_err, tol, nmin, nmax, capacity, new_mesh_points, new_nodes_per_interval, new_n_intervals,
                     interval_error, no_more_refine);
  RPM_GUARD_END(e)
}

int rpm_ph_refine_mesh(rpm_engine* h, int phase, const double* x, double tol, int nmin, int nmax, int capacity,
                       double* new_mesh_points, int* new_nodes_per_interval, int* new_n_intervals,
                       double* interval_error, int* no_more_refine) {
  if (!h) return RPM_E_INVALID;
  Engine& e = h->e;
  RPM_GUARD_BEGIN
  if (phase < 0 || phase >= e.P) return fail(e, RPM_E_INVALID, "The phase index is out of rang");
  if (e.n_instances != 1) return fail(e, RPM_E_UNSUPPORTED, "ph_refine_mesh: one instance per engine");
  if (!(tol > 0) || nmin < 2 || nmax < nmin) return fail(e, RPM_E_INVALID, "ph_refine_mesh: need tol > 0 and 2 <= Nmin <= Nmax");
  if (!x) {
    if (!e.has_solution) return fail(e, RPM_E_INVALID, "ph_refine_mesh: no x given and no solution stored");
    x = e.sol_x.data();
  }
  const rpm::PhaseHost& p = e.ph[phase];
  std::vector<double> rel(size_t(p.N + p.K + 1) * p.nx);
  int rc = rpm::dev_solution_error(e, phase, x, rel.data());
  if (rc) return rc;
  return refine_from(e, phase, rel.data(), tol, nmin, nmax, capacity, new_mesh_points, new_nodes_per_interval,
                     new_n_intervals, interval_error, no_more_refine);
  RPM_GUARD_END(e)
}

// ---- hp-Liu mesh refinement (second method) ------------------------------------
struct rpm_hpliu {
  rpm::HpLiu h;
  int n_phases = 0;
  std::string err;
};

int rpm_hpliu_create(int n_phases, double tol, int nmax, double ratio_r, rpm_hpliu** out) {
  if (!out || n_phases < 1 || !(tol > 0) || nmax < 2 || !(ratio_r > 0)) return RPM_E_INVALID;
  rpm_hpliu* p = new (std::nothrow) rpm_hpliu();
  if (!p) return RPM_E_INVALID;
  p->n_phases = n_phases;
  p->h.tol = tol;
  p->h.Nmax = nmax;
  p->h.R = ratio_r;
  *out = p;
  return RPM_OK;
}

void rpm_hpliu_destroy(rpm_hpliu* p) { delete p; }

const char* rpm_hpliu_last_error(const rpm_hpliu* p) { return p ? p->err.c_str() : "null hp-Liu object"; }

int rpm_hpliu_refine(rpm_hpliu* p, rpm_engine* h, const double* x, const double* rel_err, int capacity,
                     double* new_mesh_points, int* new_nodes_per_interval, int* mesh_off, int* nodes_off,
                     int* new_n_intervals, int* no_more_refine) {
  if (!p || !h) return RPM_E_INVALID;
  Engine& e = h->e;
  RPM_GUARD_BEGIN
  if (e.P != p->n_phases) return fail(e, RPM_E_INVALID, "hpliu_refine: phase count differs from rpm_hpliu_create");
  if (e.n_instances != 1) return fail(e, RPM_E_UNSUPPORTED, "hpliu_refine: one instance per engine");
  if (!x) {
    if (!e.has_solution) return fail(e, RPM_E_INVALID, "hpliu_refine: no x given and no solution stored");
    x = e.sol_x.data();
  }
  std::vector<std::vector<double>> rel(e.P);
  const double* src = rel_err;
  for (int ip = 0; ip < e.P; ++ip) {
    const rpm::PhaseHost& ph = e.ph[ip];
    const size_t cnt = size_t(ph.N + ph.K + 1) * ph.nx;
    rel[ip].resize(cnt);
    if (src) {   // the caller's matrices, phase after phase (host only)
      std::memcpy(rel[ip].data(), src, cnt * sizeof(double));
      src += cnt;
    } else {
      int rc = rpm::dev_solution_error(e, ip, x, rel[ip].data());
      if (rc) return rc;
    }
  }
  std::vector<std::vector<double>> mesh;
  std::vector<std::vector<int>> nodes;
  bool done = false;
  int rc = p->h.refine(e, x, rel, mesh, nodes, &done, &p->err);
  if (rc) return fail(e, rc, p->err.c_str());
  int moff = 0, noff = 0;
  for (int ip = 0; ip < e.P; ++ip) {
    const int nk = int(nodes[ip].size());
    if (moff + nk + 1 > capacity || noff + nk > capacity) return fail(e, RPM_E_INVALID, "hpliu_refine: capacity too small for the new meshes (the refinement itself has been recorded)");
    std::memcpy(new_mesh_points + moff, mesh[ip].data(), sizeof(double) * (nk + 1));
    std::memcpy(new_nodes_per_interval + noff, nodes[ip].data(), sizeof(int) * nk);
    mesh_off[ip] = moff;
    nodes_off[ip] = noff;
    new_n_intervals[ip] = nk;
    moff += nk + 1;
    noff += nk;
  }
  if (no_more_refine) *no_more_refine = done ? 1 : 0;
  return RPM_OK;
  RPM_GUARD_END(e)
}

int rpm_final_result_save(rpm_engine* h, const char* dir) {
  if (!h) return RPM_E_INVALID;
  Engine& e = h->e;
  RPM_GUARD_BEGIN
  if (!e.has_solution) return fail(e, RPM_E_INVALID, "final_result_save: no solution stored (call rpm_finalize_solution first)");
  const std::string base = (dir && *dir) ? std::string(dir) + "/" : std::string();
  for (int ip = 0; ip < e.P; ++ip) {
    const rpm::PhaseHost& p = e.ph[ip];
    const int M = p.N + 1;
    std::vector<double> t(M), st(size_t(M) * p.nx), ct(size_t(M) * (p.nu ? p.nu : 1)), cs(size_t(M) * p.nx), ham(M);
    int rc = rpm::dev_nlp2op(e, ip, e.sol_x.data(), e.sol_lambda.data(), t.data(), st.data(), ct.data(), cs.data(), nullptr,
                             ham.data(), nullptr, nullptr);
    if (rc) return rc;
    // Armadillo raw_ascii: one row per line, scientific notation (arma::diskio::save_raw_ascii)
    auto save = [&](const char* stem, const double* a, int rows, int cols) -> bool {
      FILE* f = std::fopen((base + stem + std::to_string(ip + 1)).c_str(), "w");
      if (!f) return false;
      for (int r = 0; r < rows; ++r) {
        for (int c = 0; c < cols; ++c) std::fprintf(f, " %24.16e", a[r + size_t(c) * rows]);
        std::fputc('\n', f);
      }
      std::fclose(f);
      return true;
    };
    if (!save("time", t.data(), M, 1) || !save("state", st.data(), M, p.nx) || !save("control", ct.data(), M, p.nu) ||
        !save("parameter", e.sol_x.data() + p.var0 + p.nx * (p.N + 1) + p.nu * p.N + 2, p.nq, 1) ||   // result_data_i->parameter, Nlp2OPConverter.cpp:153,213-214
 !save("costate", cs.data(), M, p.nx) || !save("Hamiltonian", ham.data(), M, 1))
      return fail(e, RPM_E_INVALID, "final_result_save: cannot write the result files");
  }
  return RPM_OK;
  RPM_GUARD_END(e)
}

// ---- device-resident variants ----------------------------------------------------------------
int rpm_eval_g_dev(rpm_engine* h, const double* d_x, double* d_g, void* stream) {
  if (!h) return RPM_E_INVALID;
  RPM_GUARD_BEGIN
  if (!d_x || !d_g) return fail(h->e, RPM_E_INVALID, "eval_g_dev: NULL pointer");
  return rpm::dev_eval_cons(h->e, d_x, d_g, nullptr, 1 | 4, stream);
  RPM_GUARD_END(h->e)
}
int rpm_eval_jac_g_dev(rpm_engine* h, const double* d_x, double* d_values, void* stream) {
  if (!h) return RPM_E_INVALID;
  RPM_GUARD_BEGIN
  if (!d_x || !d_values) return fail(h->e, RPM_E_INVALID, "eval_jac_g_dev: NULL pointer");
  return rpm::dev_eval_cons(h->e, d_x, nullptr, d_values, 2 | 4 | (h->e.opt_persistent_values ? 16 : 0), stream);
  RPM_GUARD_END(h->e)
}
int rpm_eval_pair_dev(rpm_engine* h, const double* d_x, double* d_g, double* d_values, void* stream) {
  if (!h) return RPM_E_INVALID;
  RPM_GUARD_BEGIN
  if (!d_x || !d_g || !d_values) return fail(h->e, RPM_E_INVALID, "eval_pair_dev: NULL pointer");
  return rpm::dev_eval_cons(h->e, d_x, d_g, d_values, 3 | 4 | (h->e.opt_persistent_values ? 16 : 0), stream);
  RPM_GUARD_END(h->e)
}
int rpm_eval_f_dev(rpm_engine* h, const double* d_x, double* d_obj, void* stream) {
  if (!h) return RPM_E_INVALID;
  RPM_GUARD_BEGIN
  if (!d_x || !d_obj) return fail(h->e, RPM_E_INVALID, "eval_f_dev: NULL pointer");
  return rpm::dev_eval_obj(h->e, d_x, d_obj, nullptr, stream);
  RPM_GUARD_END(h->e)
}
int rpm_eval_grad_f_dev(rpm_engine* h, const double* d_x, double* d_grad_f, void* stream) {
  if (!h) return RPM_E_INVALID;
  RPM_GUARD_BEGIN
  if (!d_x || !d_grad_f) return fail(h->e, RPM_E_INVALID, "eval_grad_f_dev: NULL pointer");
  return rpm::dev_eval_obj(h->e, d_x, nullptr, d_grad_f, stream);
  RPM_GUARD_END(h->e)
}
int rpm_eval_h_dev(rpm_engine* h, const double* d_x, double obj_factor, const double* d_lambda, double* d_values,
                   void* stream) {
  if (!h) return RPM_E_INVALID;
  RPM_GUARD_BEGIN
  if (h->e.hessian_mode != RPM_HESSIAN_EXACT) return fail(h->e, RPM_E_UNSUPPORTED, "eval_h_dev: the engine was created with hessian-approximation=limited-memory");
  if (!d_x || !d_lambda || !d_values) return fail(h->e, RPM_E_INVALID, "eval_h_dev: NULL pointer");
  return rpm::dev_eval_h(h->e, d_x, obj_factor, d_lambda, d_values, stream);
  RPM_GUARD_END(h->e)
}
int rpm_synchronize(rpm_engine* h) {
  if (!h) return RPM_E_INVALID;
  return rpm::dev_sync(h->e);
}

// ---- per-instance problem constants (parameter sweeps) -------------------------------------------
int rpm_set_instance_constants(rpm_engine* h, int instance, const double* consts, int n) {
  if (!h || !consts) return RPM_E_INVALID;
  RPM_GUARD_BEGIN
  Engine& e = h->e;
  if (instance < 0 || instance >= e.n_instances) return fail(e, RPM_E_INVALID, "rpm_set_instance_constants: instance out of range");
  if (n != int(e.consts.size())) return fail(e, RPM_E_INVALID, ("rpm_set_instance_constants: the problem functor takes " + std::to_string(e.consts.size()) + " constants").c_str());
  if (e.shard_world > 1) return fail(e, RPM_E_UNSUPPORTED, "rpm_set_instance_constants: not with interval sharding");
  if (n == 0) return RPM_OK;
  if (e.inst_consts.empty()) {
    e.inst_consts.resize(size_t(e.n_instances) * n);
    for (int b = 0; b < e.n_instances; ++b) std::copy(e.consts.begin(), e.consts.end(), e.inst_consts.begin() + size_t(b) * n);
  }
  std::copy(consts, consts + n, e.inst_consts.begin() + size_t(instance) * n);
  if (instance == 0) std::copy(consts, consts + n, e.consts.begin());   // what the one-instance (post-solve) entry points use
  return rpm::dev_update_instance_constants(e);
  RPM_GUARD_END(h->e)
}

// ---- options ----------------------------------------------------------------------------------
int rpm_set_option(rpm_engine* h, const char* key, int value) {
  if (!h || !key) return RPM_E_INVALID;
  Engine& e = h->e;
  const std::string k(key);
  if (k == "fuse_pair") e.opt_fuse_pair = value ? 1 : 0;
  else if (k == "check_finite") e.opt_check_finite = value ? 1 : 0;
  else if (k == "pin_host") {
    e.opt_pin_host = value ? 1 : 0;
    e.pin_refused.clear();
    if (!value) {   // a caller that is about to free or unmap its arrays turns the option off first
      if (e.dev) { int rc = rpm::dev_sync(e); if (rc) return rc; }
      rpm::dev_pin_release_all(e);
    }
  }
  else if (k == "dx_mode") {
    if (value != 0 && value != 1) return fail(e, RPM_E_INVALID, "dx_mode must be 0 (scalar, reference order) or 1 (MFMA)");
    if (value == 1 && e.first_derive == RPM_DERIVE_ANALYTIC)
      return fail(e, RPM_E_UNSUPPORTED, "dx_mode=1 is only built for first-derive=finite-difference");
    e.opt_dx_mode = value;
  } else if (k == "tile_nodes") {
    if (value != 0 && value != 16 && value != 32 && value != 64) return fail(e, RPM_E_INVALID, "tile_nodes must be 0, 16, 32 or 64");
    if (e.dev) return fail(e, RPM_E_INVALID, "tile_nodes must be set before the device is initialised");
    e.opt_tile_nodes = value;
    if (value) {
      e.role_looped = false;
      rpm::build_tiles(e, value);
    }
  } else if (k == "instance_align") {
    if (value < 1 || value > 65536 || (value & (value - 1))) return fail(e, RPM_E_INVALID, "instance_align must be a power of two between 1 and 65536 doubles");
    if (e.ipm_attached > 0 && value != e.opt_instance_align)
      return fail(e, RPM_E_INVALID, "instance_align cannot change while an rpm_ipm solver is attached to the engine (its buffers are sized from the strides)");
    e.opt_instance_align = value;
  } else if (k == "const_once") {
    if (value != 0 && value != 1) return fail(e, RPM_E_INVALID, "const_once must be 0 or 1");
    e.opt_const_once = value;
    e.const_filled = nullptr;
  } else if (k == "ipm_local_border") {
    if (e.ipm_attached > 0) return fail(e, RPM_E_INVALID, "ipm_local_border must be set before rpm_ipm_create");
    e.opt_ipm_local_border = value != 0.0;
  } else if (k == "ipm_nested_group") {
    if (value < 0) return fail(e, RPM_E_INVALID, "ipm_nested_group must be >= 0 (0 = automatic)");
    if (e.ipm_attached > 0) return fail(e, RPM_E_INVALID, "ipm_nested_group must be set before rpm_ipm_create");
    e.opt_ipm_nested_group = value;
  } else if (k == "ipm_nested") {
    if (value < -1 || value > 1) return fail(e, RPM_E_INVALID, "ipm_nested must be -1 (automatic), 0 or 1");
    if (e.ipm_attached > 0) return fail(e, RPM_E_INVALID, "ipm_nested must be set before rpm_ipm_create");
    e.opt_ipm_nested = value;
  } else if (k == "persistent_values") {
    if (value != 0 && value != 1) return fail(e, RPM_E_INVALID, "persistent_values must be 0 or 1");
    e.opt_persistent_values = value;
  } else if (k == "delta_values") {
    if (value != 0 && value != 1) return fail(e, RPM_E_INVALID, "delta_values must be 0 or 1");
    e.opt_delta_values = value;
  } else if (k == "zero_copy") {
    if (value != 0 && value != 1) return fail(e, RPM_E_INVALID, "zero_copy must be 0 or 1");
    e.opt_zero_copy = value;
  } else if (k == "pipeline") {
    if (value < -1 || value > 1) return fail(e, RPM_E_INVALID, "pipeline must be -1 (auto), 0 or 1");
    e.opt_pipeline = value;
  } else if (k == "stage_roles") {
    if (value < -1 || value > 1) return fail(e, RPM_E_INVALID, "stage_roles must be -1 (auto), 0 or 1");
    e.opt_stage_roles = value;
  } else if (k == "role_loop") {
    if (value < -1 || value > 1) return fail(e, RPM_E_INVALID, "role_loop must be -1 (auto), 0 or 1");
    if (e.dev) return fail(e, RPM_E_INVALID, "role_loop must be set before the device is initialised");
    e.opt_role_loop = value;
    if (e.opt_tile_nodes == 0) {   // re-tile with the new policy
      long long total = 0;
      for (int i = 0; i < e.P; ++i) total += e.ph[i].N;
      total *= e.n_instances;
      e.role_looped = value == 1 || (value == -1 && total / 64 >= 512);
      rpm::build_tiles(e, e.role_looped ? 64 : 16);
    }
  } else
    return fail(e, RPM_E_INVALID, "unknown option");
  rpm::host_new_x(e);   // nothing cached under the old options is handed out under the new ones
  rpm::dev_forget_persistent(e);
  return RPM_OK;
}
int rpm_get_option(rpm_engine* h, const char* key, int* value) {
  if (!h || !key || !value) return RPM_E_INVALID;
  Engine& e = h->e;
  const std::string k(key);
  if (k == "fuse_pair") *value = e.opt_fuse_pair;
  else if (k == "check_finite") *value = e.opt_check_finite;
  else if (k == "dx_mode") *value = e.opt_dx_mode;
  else if (k == "tile_nodes") *value = e.tile_nodes;
  else if (k == "n_tiles") *value = int(e.tiles.size());
  else if (k == "role_loop") *value = e.role_looped ? 1 : 0;
  else if (k == "pipeline") *value = e.opt_pipeline;
  else if (k == "stage_roles") *value = e.opt_stage_roles;
  else if (k == "const_once") *value = e.opt_const_once;
  else if (k == "instance_align") *value = e.opt_instance_align;
  else if (k == "delta_values") *value = e.opt_delta_values;
  else if (k == "ipm_nested") *value = e.opt_ipm_nested;
  else if (k == "ipm_nested_group") *value = e.opt_ipm_nested_group;
  else if (k == "ipm_local_border") *value = e.opt_ipm_local_border;
  else if (k == "zero_copy") *value = e.opt_zero_copy;
  else if (k == "persistent_values") *value = e.opt_persistent_values;
  else if (k == "pin_host") *value = e.opt_pin_host;
  // page-lock registry of the process (librpm_pin.so; counters since the process started) and this engine's share of it
  else if (k == "pin_registered") *value = int(rpm::dev_pin_counter(RPM_PIN_REGISTERED));
  else if (k == "pin_register_failures") *value = int(rpm::dev_pin_counter(RPM_PIN_REGISTER_FAILURES));
  else if (k == "pin_unregistered") *value = int(rpm::dev_pin_counter(RPM_PIN_UNREGISTERED));
  else if (k == "pin_unregister_failures") *value = int(rpm::dev_pin_counter(RPM_PIN_UNREGISTER_FAILURES));
  else if (k == "pin_overlap_refused") *value = int(rpm::dev_pin_counter(RPM_PIN_OVERLAP_REFUSED));
  else if (k == "pin_shared") *value = int(rpm::dev_pin_counter(RPM_PIN_SHARED));
  else if (k == "pin_merged") *value = int(rpm::dev_pin_counter(RPM_PIN_MERGED));
  else if (k == "pin_evicted") *value = int(rpm::dev_pin_counter(RPM_PIN_EVICTED));
  else if (k == "pin_live") *value = int(rpm::dev_pin_counter(RPM_PIN_LIVE));
  else if (k == "pin_live_kb") *value = int(rpm::dev_pin_counter(RPM_PIN_LIVE_BYTES) / 1024);
  else if (k == "pin_held") *value = rpm::dev_pin_held(e);
  else if (k == "delta_sent_runs") return rpm::host_delta_sent_runs(e, value);
  else if (k == "delta_total_runs") *value = e.last_delta_total;
  else if (k == "stride_g") *value = int(e.stride_g());
  else if (k == "stride_values") *value = int(e.stride_values());
  else if (k == "pipeline_active") *value = rpm::dev_pipeline_active(e);
  else return fail(e, RPM_E_INVALID, "unknown option");
  return RPM_OK;
}

// ---- collocation tables ---------------------------------------------------------------------
int rpm_get_phase_sizes(rpm_engine* h, int phase, int* n_nodes, int* d_nnz, int* doff_nnz) {
  if (!h) return RPM_E_INVALID;
  Engine& e = h->e;
  if (phase < 0 || phase >= e.P) return fail(e, RPM_E_INVALID, "The phase index is out of rang");
  if (n_nodes) *n_nodes = e.ph[phase].N;
  if (d_nnz) *d_nnz = int(e.ph[phase].d_v.size());
  if (doff_nnz) *doff_nnz = int(e.ph[phase].off_v.size());
  return RPM_OK;
}
int rpm_get_phase_tables(rpm_engine* h, int phase, double* points, double* weights, int* d_rows, int* d_cols,
                         double* d_vals, double* diag_vals, int* doff_rows, int* doff_cols, double* doff_vals) {
  if (!h) return RPM_E_INVALID;
  Engine& e = h->e;
  if (phase < 0 || phase >= e.P) return fail(e, RPM_E_INVALID, "The phase index is out of rang");
  const rpm::PhaseHost& p = e.ph[phase];
  auto cp = [](auto* dst, const auto& v) {
    if (dst && !v.empty()) std::memcpy(dst, v.data(), v.size() * sizeof(v[0]));
  };
  cp(points, p.points);
  cp(weights, p.weights);
  cp(d_rows, p.d_i);
  cp(d_cols, p.d_j);
  cp(d_vals, p.d_v);
  cp(diag_vals, p.diag_v);
  cp(doff_rows, p.off_i);
  cp(doff_cols, p.off_j);
  cp(doff_vals, p.off_v);
  return RPM_OK;
}

// ---- interval sharding ---------------------------------------------------------------------------
int rpm_shard_segments(rpm_engine* h, int which, int rank, rpm_segment* seg, int* n_seg, int* packed_len) {
  if (!h) return RPM_E_INVALID;
  Engine& e = h->e;
  RPM_GUARD_BEGIN
  if (which < 0 || which > 1 || rank < 0 || rank >= e.shard_world) return fail(e, RPM_E_INVALID, "shard_segments: bad argument");
  int plen = 0;
  std::vector<rpm_segment> s = rpm::shard_segments(e, which, rank, &plen);
  if (seg) {
    if (!n_seg || *n_seg < int(s.size())) return fail(e, RPM_E_INVALID, "shard_segments: segment buffer too small");
    std::memcpy(seg, s.data(), s.size() * sizeof(rpm_segment));
  }
  if (n_seg) *n_seg = int(s.size());
  if (packed_len) *packed_len = plen;
  return RPM_OK;
  RPM_GUARD_END(e)
}
int rpm_shard_pack_dev(rpm_engine* h, int which, const double* d_full, double* d_packed, void* stream) {
  if (!h) return RPM_E_INVALID;
  RPM_GUARD_BEGIN
  if (!d_full || !d_packed || which < 0 || which > 1) return fail(h->e, RPM_E_INVALID, "shard_pack_dev: bad argument");
  return rpm::dev_shard_copy(h->e, which, true, d_full, 0, d_packed, stream);
  RPM_GUARD_END(h->e)
}
int rpm_shard_unpack_dev(rpm_engine* h, int which, const double* d_gathered, int stride, double* d_full, void* stream) {
  if (!h) return RPM_E_INVALID;
  RPM_GUARD_BEGIN
  if (!d_gathered || !d_full || which < 0 || which > 1) return fail(h->e, RPM_E_INVALID, "shard_unpack_dev: bad argument");
  return rpm::dev_shard_copy(h->e, which, false, d_gathered, stride, d_full, stream);
  RPM_GUARD_END(h->e)
}
int rpm_shard_slot_len(rpm_engine* h, long long* slot_doubles) {
  if (!h || !slot_doubles) return RPM_E_INVALID;
  RPM_GUARD_BEGIN
  *slot_doubles = rpm::shard_slot_len(h->e);
  return RPM_OK;
  RPM_GUARD_END(h->e)
}
int rpm_shard_pack_all_dev(rpm_engine* h, const double* d_g, const double* d_values, double* d_slot, void* stream) {
  if (!h) return RPM_E_INVALID;
  RPM_GUARD_BEGIN
  if (!d_g || !d_values || !d_slot) return fail(h->e, RPM_E_INVALID, "shard_pack_all_dev: NULL pointer");
  return rpm::dev_shard_pack_all(h->e, d_g, d_values, d_slot, stream);
  RPM_GUARD_END(h->e)
}
int rpm_shard_unpack_all_dev(rpm_engine* h, const double* d_gathered, double* d_g, double* d_values, int skip_own, void* stream) {
  if (!h) return RPM_E_INVALID;
  RPM_GUARD_BEGIN
  if (!d_g || !d_values || !d_gathered) return fail(h->e, RPM_E_INVALID, "shard_unpack_all_dev: NULL pointer");
  return rpm::dev_shard_unpack_all(h->e, d_gathered, d_g, d_values, skip_own, stream);
  RPM_GUARD_END(h->e)
}
}  // extern "C"

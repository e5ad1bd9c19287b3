// RpmTNLP.hpp — the adaptor that takes LpopcIpopt's place (Core/LpopcIpopt.h:18-104): an
// Ipopt::TNLP subclass whose eight virtuals forward to the C ABI of librpm_hip.so.
//
// It is a template over the TNLP base so that it can be compiled (and unit-tested) without
// Ipopt's headers; with Ipopt available use
//     #include <IpTNLP.hpp>
//     using RpmTNLP = lpopc_amd::RpmTNLPT<Ipopt::TNLP>;
// `Base` must provide the typedefs Index, Number, IndexStyleEnum (with C_STYLE), SolverReturn and the
// forward declarations IpoptData / IpoptCalculatedQuantities used by finalize_solution, exactly as
// Ipopt::TNLP does.
#pragma once
#include <string>

#include "../rpm_hip.h"

namespace lpopc_amd {

template <class Base>
class RpmTNLPT : public Base {
 public:
  using Index = typename Base::Index;
  using Number = typename Base::Number;
  using IndexStyleEnum = typename Base::IndexStyleEnum;
  using SolverReturn = typename Base::SolverReturn;

  // ipopt_owned_arrays = true (what Ipopt's TNLPAdapter guarantees: it allocates full_x_, full_g_ and jac_g_ once per
  // solve, hands the same arrays to every callback and never writes into jac_g_ itself): the engine page-locks them on
  // first sight ("pin_host"), reads x / stores g in place ("zero_copy") and delivers `values` by difference
  // ("delta_values": the constant Doffdiag block, the linear entries and every x-independent block cross PCIe once; the
  // engine still checks 64 sampled entries of the array before each delivery and re-sends everything if one differs).
  // Pass false for a caller that allocates fresh arrays per call or edits `values` in place: plain staged copies.
  // ("pin_host" is off in the C ABI unless asked for: this constructor argument is the asking.)
  explicit RpmTNLPT(rpm_engine* engine, bool ipopt_owned_arrays = true) : e_(engine), owned_(ipopt_owned_arrays) {
    rpm_set_option(e_, "pin_host", owned_ ? 1 : 0);
    rpm_set_option(e_, "delta_values", owned_ ? 1 : 0);
  }

  bool get_nlp_info(Index& n, Index& m, Index& nnz_jac_g, Index& nnz_h_lag, IndexStyleEnum& index_style) override {
    int n_, m_, nj, nh, st;
    if (rpm_get_nlp_info(e_, &n_, &m_, &nj, &nh, &st)) return false;
    n = n_; m = m_; nnz_jac_g = nj; nnz_h_lag = nh;
    index_style = Base::C_STYLE;                                 // LpopcIpopt.cpp:22
    return true;
  }
  bool get_bounds_info(Index n, Number* x_l, Number* x_u, Index m, Number* g_l, Number* g_u) override {
    return rpm_get_bounds_info(e_, n, x_l, x_u, m, g_l, g_u) == RPM_OK;
  }
  bool get_starting_point(Index n, bool init_x, Number* x, bool init_z, Number* z_L, Number* z_U, Index m,
                          bool init_lambda, Number* lambda) override {
    return rpm_get_starting_point(e_, n, init_x, x, init_z, z_L, z_U, m, init_lambda, lambda) == RPM_OK;
  }
  bool eval_f(Index n, const Number* x, bool new_x, Number& obj_value) override {
    return rpm_eval_f(e_, n, x, new_x, &obj_value) == RPM_OK;
  }
  bool eval_grad_f(Index n, const Number* x, bool new_x, Number* grad_f) override {
    return rpm_eval_grad_f(e_, n, x, new_x, grad_f) == RPM_OK;
  }
  bool eval_g(Index n, const Number* x, bool new_x, Index m, Number* g) override {
    return rpm_eval_g(e_, n, x, new_x, m, g) == RPM_OK;
  }
  bool eval_jac_g(Index n, const Number* x, bool new_x, Index m, Index nele_jac, Index* iRow, Index* jCol,
                  Number* values) override {
    return rpm_eval_jac_g(e_, n, x, new_x, m, nele_jac, iRow, jCol, values) == RPM_OK;
  }
  bool eval_h(Index n, const Number* x, bool new_x, Number obj_factor, Index m, const Number* lambda, bool new_lambda,
              Index nele_hess, Index* iRow, Index* jCol, Number* values) override {
    return rpm_eval_h(e_, n, x, new_x, obj_factor, m, lambda, new_lambda, nele_hess, iRow, jCol, values) == RPM_OK;
  }
  void finalize_solution(SolverReturn status, Index n, const Number* x, const Number* z_L, const Number* z_U, Index m,
                         const Number* g, const Number* lambda, Number obj_value,
                         const typename Base::IpoptData* /*ip_data*/,
                         typename Base::IpoptCalculatedQuantities* /*ip_cq*/) override {
    rpm_finalize_solution(e_, int(status), n, x, z_L, z_U, m, g, lambda, obj_value);  // LpopcIpopt.cpp:220-246
    // Ipopt's last call into the TNLP, and its x / g / values arrays (TNLPAdapter's) may be freed before this object is:
    // release their page-locked registrations now (rpm_hip.h "pin_host": a registered array must not be unmapped), and
    // put the option back to what the constructor was told, for a further OptimizeTNLP with the same object.
    rpm_set_option(e_, "pin_host", 0);
    rpm_set_option(e_, "pin_host", owned_ ? 1 : 0);
  }
  std::string last_error() const { return rpm_last_error(e_); }

 private:
  rpm_engine* e_;
  bool owned_;   // the caller's promise about the arrays (constructor)
};

// The same adaptor over a group of engines, one per GPU (rpm_group_*, include/rpm_hip.h): lpopc's NLPSolver::SolveNlp is one
// process driving ONE TNLP object, so this is how its callbacks reach more than one device.  The constraint callbacks go to
// every device (each stores its share into Ipopt's arrays, which a group page-locks once for all devices); objective,
// gradient and exact Hessian to rank 0; sizes, bounds, starting point and the stored solution to rank 0's engine.
template <class Base>
class RpmGroupTNLPT : public Base {
 public:
  using Index = typename Base::Index;
  using Number = typename Base::Number;
  using IndexStyleEnum = typename Base::IndexStyleEnum;
  using SolverReturn = typename Base::SolverReturn;

  explicit RpmGroupTNLPT(rpm_group* group) : g_(group), e0_(rpm_group_engine(group, 0)) {}

  bool get_nlp_info(Index& n, Index& m, Index& nnz_jac_g, Index& nnz_h_lag, IndexStyleEnum& index_style) override {
    int n_, m_, nj, nh, st;
    if (rpm_get_nlp_info(e0_, &n_, &m_, &nj, &nh, &st)) return false;
    n = n_; m = m_; nnz_jac_g = nj; nnz_h_lag = nh;
    index_style = Base::C_STYLE;
    return true;
  }
  bool get_bounds_info(Index n, Number* x_l, Number* x_u, Index m, Number* g_l, Number* g_u) override {
    return rpm_get_bounds_info(e0_, n, x_l, x_u, m, g_l, g_u) == RPM_OK;
  }
  bool get_starting_point(Index n, bool init_x, Number* x, bool init_z, Number* z_L, Number* z_U, Index m,
                          bool init_lambda, Number* lambda) override {
    return rpm_get_starting_point(e0_, n, init_x, x, init_z, z_L, z_U, m, init_lambda, lambda) == RPM_OK;
  }
  bool eval_f(Index n, const Number* x, bool new_x, Number& obj_value) override {
    return rpm_group_eval_f(g_, n, x, new_x, &obj_value) == RPM_OK;
  }
  bool eval_grad_f(Index n, const Number* x, bool new_x, Number* grad_f) override {
    return rpm_group_eval_grad_f(g_, n, x, new_x, grad_f) == RPM_OK;
  }
  bool eval_g(Index n, const Number* x, bool new_x, Index m, Number* g) override {
    return rpm_group_eval_g(g_, n, x, new_x, m, g) == RPM_OK;
  }
  bool eval_jac_g(Index n, const Number* x, bool new_x, Index m, Index nele_jac, Index* iRow, Index* jCol,
                  Number* values) override {
    return rpm_group_eval_jac_g(g_, n, x, new_x, m, nele_jac, iRow, jCol, values) == RPM_OK;
  }
  bool eval_h(Index n, const Number* x, bool new_x, Number obj_factor, Index m, const Number* lambda, bool new_lambda,
              Index nele_hess, Index* iRow, Index* jCol, Number* values) override {
    return rpm_group_eval_h(g_, n, x, new_x, obj_factor, m, lambda, new_lambda, nele_hess, iRow, jCol, values) == RPM_OK;
  }
  void finalize_solution(SolverReturn status, Index n, const Number* x, const Number* z_L, const Number* z_U, Index m,
                         const Number* g, const Number* lambda, Number obj_value,
                         const typename Base::IpoptData* /*ip_data*/,
                         typename Base::IpoptCalculatedQuantities* /*ip_cq*/) override {
    rpm_finalize_solution(e0_, int(status), n, x, z_L, z_U, m, g, lambda, obj_value);   // LpopcIpopt.cpp:220-246
    // Ipopt's arrays may be freed before this object is: every engine lets go of them now; the next host-consumer call
    // of the group registers whatever arrays it is handed then
    rpm_group_set_option(g_, "pin_host", 0);
  }
  std::string last_error() const { return rpm_group_last_error(g_); }

 private:
  rpm_group* g_;
  rpm_engine* e0_;
};

}  // namespace lpopc_amd

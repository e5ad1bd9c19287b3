// OptimalProblem.hpp — header-only C++ façade with lpopc's problem-setup API (same class and method
// names as Core/LpOptimalProblem.hpp:30-326) that lowers to the C ABI's rpm_problem_desc.
// The FunctionWrapper argument of OptimalProblem is replaced by a ProblemFunctor: the id of a device
// functor compiled into librpm_hip.so plus the problem constants.
#pragma once
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../rpm_hip.h"

namespace lpopc_amd {

using std::shared_ptr;

struct LpopcException : std::runtime_error {   // Common/LpException.hpp:14
  using std::runtime_error::runtime_error;
};

struct Limit {                                  // Core/LpOptimalProblem.hpp:18-29
  Limit(double state0, double state_, double statef) : state{state0, state_, statef} {}
  void GetLimit(double& s0, double& s, double& sf) const { s0 = state[0]; s = state[1]; sf = state[2]; }
  double state[3];
};

struct ProblemFunctor {
  int problem_id;
  std::vector<double> consts;
};

class Phase {                                   // Core/LpOptimalProblem.hpp:30-240
 public:
  Phase(int phase_index, int statenum, int controlnum, int parameternum, int pathnum, int eventnum)
      : phase_index_(phase_index), statenum_(statenum), controlnum_(controlnum), parameternum_(parameternum),
        pathnum_(pathnum), eventnum_(eventnum) {}
  void get_optimal_info(int& s, int& c, int& p, int& pa, int& e) const {
    s = statenum_; c = controlnum_; p = parameternum_; pa = pathnum_; e = eventnum_;
  }
  void SetTimeMin(double t0, double tf) { tmin_[0] = t0; tmin_[1] = tf; }
  void SetTimeMax(double t0, double tf) { tmax_[0] = t0; tmax_[1] = tf; }
  void SetStateMin(double s0, double s, double sf) { smin_.insert(smin_.end(), {s0, s, sf}); }
  void SetStateMax(double s0, double s, double sf) { smax_.insert(smax_.end(), {s0, s, sf}); }
  void SetcontrolMin(double v) { cmin_.push_back(v); }
  void SetcontrolMax(double v) { cmax_.push_back(v); }
  void SetparameterlMin(double v) { pmin_.push_back(v); }   // (sic) :97
  void SetparameterMax(double v) { pmax_.push_back(v); }
  void SetpathMin(double v) { pathmin_.push_back(v); }
  void SetpathMax(double v) { pathmax_.push_back(v); }
  void SeteventMin(double v) { evmin_.push_back(v); }
  void SeteventMax(double v) { evmax_.push_back(v); }
  void SetDuration(double mn, double mx) { hasduration_ = true; dur_[0] = mn; dur_[1] = mx; }
  void SetTimeGuess(double g) { tguess_.push_back(g); }
  void SetStateGuess(int stateindex, double g) {            // 1-based, :135-143
    if ((int)xguess_.size() >= stateindex) xguess_[stateindex - 1].push_back(g);
    else if (stateindex == (int)xguess_.size() + 1) xguess_.push_back({g});
  }
  void SetControlGuess(int controlindex, double g) {
    if ((int)uguess_.size() >= controlindex) uguess_[controlindex - 1].push_back(g);
    else if (controlindex == (int)uguess_.size() + 1) uguess_.push_back({g});
  }
  void SetparameterGuess(double g) { pguess_.push_back(g); }
  void SetMeshPoints(double m) { mesh_.push_back(m); }
  void SetNodesPerInterval(int n) { nodes_.push_back(n); }
  bool HasDuration() const { return hasduration_; }

  // lowering: fills `d` with pointers into this object (which must outlive the descriptor)
  void Lower(rpm_phase_desc& d) {
    // MeshRefiner::SetAndCheckMesh defaults, Core/LpMeshRefiner.cpp:10-62
    if (mesh_.empty()) {
      const size_t k = nodes_.empty() ? 1 : nodes_.size();
      for (size_t i = 0; i <= k; ++i) mesh_.push_back(i == k ? 1.0 : -1.0 + 2.0 * double(i) / double(k));
    }
    if (nodes_.empty()) nodes_.assign(mesh_.size() - 1, 20);
    if (mesh_.size() != nodes_.size() + 1)
      throw LpopcException("Number of nodesPerInterval must match number of mesh intervals in phase" +
                           std::to_string(phase_index_));
    flat_x_.clear();
    for (auto& r : xguess_) flat_x_.insert(flat_x_.end(), r.begin(), r.end());
    flat_u_.clear();
    for (auto& r : uguess_) flat_u_.insert(flat_u_.end(), r.begin(), r.end());
    d = rpm_phase_desc{};
    d.nx = statenum_; d.nu = controlnum_; d.nq = parameternum_; d.nc = pathnum_; d.ne = eventnum_;
    d.n_intervals = int(nodes_.size());
    d.mesh_points = mesh_.data();
    d.nodes_per_interval = nodes_.data();
    d.t0_min = tmin_[0]; d.tf_min = tmin_[1]; d.t0_max = tmax_[0]; d.tf_max = tmax_[1];
    d.state_min = smin_.data(); d.state_max = smax_.data();
    d.control_min = cmin_.data(); d.control_max = cmax_.data();
    d.parameter_min = pmin_.data(); d.parameter_max = pmax_.data();
    d.path_min = pathmin_.data(); d.path_max = pathmax_.data();
    d.event_min = evmin_.data(); d.event_max = evmax_.data();
    d.has_duration = hasduration_ ? 1 : 0; d.duration_min = dur_[0]; d.duration_max = dur_[1];
    d.n_guess = int(tguess_.size());
    d.time_guess = tguess_.data(); d.state_guess = flat_x_.data(); d.control_guess = flat_u_.data();
    d.parameter_guess = pguess_.data();
  }

 private:
  int phase_index_, statenum_, controlnum_, parameternum_, pathnum_, eventnum_;
  bool hasduration_ = false;
  double tmin_[2] = {0, 0}, tmax_[2] = {0, 0}, dur_[2] = {0, 0};
  std::vector<double> smin_, smax_, cmin_, cmax_, pmin_, pmax_, pathmin_, pathmax_, evmin_, evmax_;
  std::vector<double> tguess_, pguess_, mesh_, flat_x_, flat_u_;
  std::vector<std::vector<double>> xguess_, uguess_;
  std::vector<int> nodes_;
};

class Linkage {                                 // Core/LpOptimalProblem.hpp:242-279
 public:
  Linkage(int ipair, int left, int right) : pairindex(ipair), leftphase(left), rightphase(right) {}
  void SetLinkMin(double v) { linkmin.push_back(v); }
  void SetLinkMax(double v) { linkmax.push_back(v); }
  int LeftPhase() const { return leftphase - 1; }
  int RightPhase() const { return rightphase - 1; }
  void Lower(rpm_link_desc& d) const {
    d.left_phase = leftphase; d.right_phase = rightphase; d.n_links = int(linkmin.size());
    d.link_min = linkmin.data(); d.link_max = linkmax.data();
  }
 private:
  int pairindex, leftphase, rightphase;
  std::vector<double> linkmin, linkmax;
};

class OptimalProblem {                          // Core/LpOptimalProblem.hpp:281-326
 public:
  OptimalProblem(int numphase, int numlinkage, shared_ptr<ProblemFunctor> userfun)
      : numphase_(numphase), numlink_(numlinkage), userfunction_(std::move(userfun)) {}
  void AddPhase(shared_ptr<Phase>& p) { Phases_.push_back(p); }
  void AddLinkage(shared_ptr<Linkage>& l) { Linkage_.push_back(l); }
  shared_ptr<Phase>& GetPhase(size_t i) {
    if (i >= Phases_.size()) throw LpopcException("The phase index is out of rang in Function 'GetPhase' ");
    return Phases_[i];
  }
  int GetPhaseNum() const { return numphase_; }
  int GetLinkageNum() const { return numlink_; }

  // the descriptor points into this object
  const rpm_problem_desc& Lower(double fd_tol = 1e-6, int first_derive = RPM_DERIVE_FINITE_DIFFERENCE,
                                int hessian = RPM_HESSIAN_LIMITED_MEMORY, int n_instances = 1) {
    if ((int)Phases_.size() != numphase_ || (int)Linkage_.size() != numlink_)
      throw LpopcException("number of phases/linkages added does not match the OptimalProblem constructor");
    pd_.resize(Phases_.size());
    ld_.resize(Linkage_.size());
    for (size_t i = 0; i < Phases_.size(); ++i) Phases_[i]->Lower(pd_[i]);
    for (size_t i = 0; i < Linkage_.size(); ++i) Linkage_[i]->Lower(ld_[i]);
    desc_ = rpm_problem_desc{};
    desc_.abi_version = RPM_ABI_VERSION;
    desc_.problem_id = userfunction_->problem_id;
    desc_.n_phases = numphase_; desc_.phases = pd_.data();
    desc_.n_links = numlink_; desc_.links = ld_.data();
    desc_.n_consts = int(userfunction_->consts.size()); desc_.consts = userfunction_->consts.data();
    desc_.fd_tol = fd_tol; desc_.first_derive = first_derive; desc_.hessian_approximation = hessian;
    desc_.n_instances = n_instances; desc_.shard_mode = RPM_SHARD_NONE; desc_.shard_rank = 0; desc_.shard_world = 1;
    return desc_;
  }
 private:
  int numphase_, numlink_;
  shared_ptr<ProblemFunctor> userfunction_;
  std::vector<shared_ptr<Phase>> Phases_;
  std::vector<shared_ptr<Linkage>> Linkage_;
  std::vector<rpm_phase_desc> pd_;
  std::vector<rpm_link_desc> ld_;
  rpm_problem_desc desc_{};
};

}  // namespace lpopc_amd

// rpm_device_internal.hpp — what the HIP translation units of the engine share: the kernel parameter blocks, the
// device-side state of an engine, the problem registry and small helpers.  Not part of the ABI.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "problems/problems.hpp"
#ifdef RPM_USER_PROBLEM_HEADER
#include RPM_USER_PROBLEM_HEADER   // defines struct rpm::UserProblem (same interface as the structs of problems.hpp)
#endif
#include "rpm_engine.hpp"

namespace rpm {

struct KParams {
  const PhaseDev* phases;
  const TileDev* tiles;  // the tiles this rank computes, compact
  int n_my_tiles;
  const TaskDev* tasks;  // endpoint work items, one extra workgroup each
  int n_tasks;
  const NodeDev* nodes;
  const double* points;
  const double* weights;
  const double* diag;
  const double* dvals;
  const double* doff_vals;
  const double* consts;
  int consts_stride;                    // doubles between the constants of consecutive instances (0: shared)
  const LinkDev* links;
  const int* alin_j;     // 2 entries per linear row
  const double* alin_v;
  double tol;
  int P, L, n, m, m_nl, nnz, nnz_nl, nnz_lin, nnz_const;
  long long sg, sv;                     // distance between the g / values arrays of consecutive instances (>= m, nnz)
  int max_span, max_drow;
  int max_cshare;                       // largest constant-block share of a tile (c_cnt)
  int skip_const;                       // 1: this launch leaves the constant Doffdiag block of `values` alone (persistent arrays, dev_eval_cons)
  int diag_mask;                        // ablation mask, only honoured by the -DRPM_DIAG diagnostic build
  unsigned long long* trace;            // per-workgroup timestamps (diagnostic build with RPM_DIAG_TRACE set), else NULL
  int* chk;                             // host-pointer path: two host-visible words ORed with "a stored g / Jacobian value is NaN/Inf"; else NULL
};

struct HParams {
  const HessPairDev* pairs;
  const HessPhaseDev* phases;
  const HessEndDev* ends;
  const HessLinkDev* links;
  const int* tiles;     // per workgroup: phase, k0, cnt
  int n_tiles, th, n_ends, n_links, nnz_h, tmp_len;
};

struct Device {
  int device_id = -1;
  int* d_flags2 = nullptr;      // two non-finite flag words (g, Jacobian) and their page-locked host mirror
  int* h_flags2 = nullptr;
  size_t trace_words = 0;
  hipStream_t stream = nullptr;
  KParams kp{};
  // tables
  PhaseDev* d_phases = nullptr;
  TileDev* d_tiles = nullptr;
  TaskDev* d_tasks = nullptr;
  NodeDev* d_nodes = nullptr;
  double* d_inst_consts = nullptr;   // n_instances x nconst when rpm_set_instance_constants was used
  double *d_points = nullptr, *d_weights = nullptr, *d_diag = nullptr, *d_dvals = nullptr,
         *d_doff = nullptr, *d_consts = nullptr, *d_alin_v = nullptr;
  LinkDev* d_links = nullptr;
  int* d_alin_j = nullptr;
  // staging buffers of the host-pointer TNLP path
  double *d_x = nullptr, *d_g = nullptr, *d_values = nullptr, *d_grad = nullptr, *d_obj = nullptr,
         *d_lambda = nullptr, *d_hess = nullptr;
  double* d_partial = nullptr;  // objective partial sums
  int* d_flag = nullptr;        // non-finite flag of the host-pointer path
  // page-locked staging buffers the engine owns (hipHostMalloc, mapped): caller arrays that are not registered
  // ("pin_host" off, small, refused) are copied through these by the CPU and never reach the HIP runtime
  struct Stage { double* h = nullptr; double* d = nullptr; size_t cap = 0; bool busy = false; };   // busy: a queued H2D copy still reads it
  Stage stage[STAGE_SLOTS];
  bool cache_valid = false;     // d_g / d_values hold the pair of the x last uploaded
  std::vector<const double*> const_filled;   // device `values` arrays whose constant block this engine has written (option "persistent_values")
  size_t lds_bytes = 0;
  int pl_slots = 0;             // resident workgroups the pipelined kernel is launched with (2 per CU)
  size_t pl_lds = 0;
  bool pl_ok = false;           // the mesh fits rpm_tile_pl_kernel's register staging
  // exact-Hessian tables
  HessPairDev* d_hpairs = nullptr;
  HessPhaseDev* d_hphases = nullptr;
  HessEndDev* d_hends = nullptr;
  HessLinkDev* d_hlinks = nullptr;
  int* d_htiles = nullptr;
  double* d_htmp = nullptr;
  HParams hp{};
  size_t hess_lds = 0;
  int hess_threads = 0;
  void* host_path = nullptr;    // state of the host-pointer delivery (rpm_host_path.hip)
  void* exchange = nullptr;     // pack / unpack tables of the interval-sharded exchange (rpm_peer.hip)
  struct SegTable { void* ptr = nullptr; int count = 0; int stride = -1; };
  SegTable segtab[2][2];        // [g|values][pack|unpack] run tables of the interval sharding
};

#define HIP_TRY(e, call)                                                                   \
  do {                                                                                     \
    hipError_t _s = (call);                                                                \
    if (_s != hipSuccess) {                                                                \
      (e).err = std::string(#call) + ": " + hipGetErrorString(_s);                         \
      return RPM_E_DEVICE;                                                                 \
    }                                                                                      \
  } while (0)

// Host-pointer path (K.chk != nullptr): every thread remembers whether a value it stored was NaN/Inf; the verdicts are
// ORed into host-visible words (word 0: g, word 1: Jacobian values, word 3: objective gradient) — no separate scan
// kernel, no extra launch.  An atomic is issued only when something is wrong.
__device__ __forceinline__ void chk_note(bool& bad, double v) { bad |= !(fabs(v) <= 1.7976931348623157e308); }
__device__ __forceinline__ void chk_report(int* chk, bool bad_g, bool bad_j) {
  if (chk == nullptr) return;
  if (bad_g) atomicOr_system(chk, 1);
  if (bad_j) atomicOr_system(chk + 1, 1);
}

// ------------------------------------------------------------------------------------------
// problem registry: calls fn with a value of the functor type registered under `id`
template <class F>
inline bool with_problem(int id, F&& fn) {
  switch (id) {
#ifndef RPM_ONLY_USER_PROBLEM   // a user library may leave the built-in functors out (6x shorter build)
    case RPM_PROBLEM_LAUNCH: fn(LaunchProblem{}); return true;
    case RPM_PROBLEM_HYPERSENSITIVE: fn(HypersensitiveProblem{}); return true;
    case RPM_PROBLEM_BRYSON_DENHAM: fn(BrysonDenhamProblem{}); return true;
    case RPM_PROBLEM_BRACHISTOCHRONE: fn(BrachistochroneProblem{}); return true;
    case RPM_PROBLEM_MIN_TIME_CLIMB: fn(MinTimeClimbProblem{}); return true;
    case RPM_PROBLEM_QUADROTOR: fn(QuadrotorProblem{}); return true;
    case RPM_PROBLEM_PARAM_SLED: fn(ParamSledProblem{}); return true;
    case RPM_PROBLEM_PARAM_OSC: fn(ParamOscProblem{}); return true;
#endif
#ifdef RPM_USER_PROBLEM_HEADER
    case RPM_PROBLEM_USER: fn(UserProblem{}); return true;
#endif
  }
  return false;
}


template <class T>
inline hipError_t upload(T** dst, const std::vector<T>& src) {
  const size_t bytes = (src.size() ? src.size() : 1) * sizeof(T);
  hipError_t s = hipMalloc(reinterpret_cast<void**>(dst), bytes);
  if (s != hipSuccess) return s;
  if (!src.empty()) s = hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice);
  return s;
}


void host_path_destroy(Device* d);   // rpm_host_path.hip
// rpm_device.hip: staging slots (see Device::Stage)
int dev_stage_reserve(Engine& e, int slot, size_t count, double** host, double** alias);
int dev_stage_upload(Engine& e, int slot, double* dev, const double* host, size_t count);
int dev_stage_download(Engine& e, int slot, double* host, const double* dev, size_t count);
void dev_stage_destroy(Device* d);
void host_new_x(Engine& e);
void dev_stage_synced(Engine& e);    // the engine's stream was synchronised: every staging slot may be overwritten
std::string dev_pin_last_error();
void exchange_destroy(Device* d);    // rpm_peer.hip

// rpm_tile_kernels.hip: occupancy, LDS size and eligibility of the pipelined kernel for this engine (device_init)
void tile_pipeline_setup(Engine& e, Device* d, const ProblemDims& pd, int device_id);

}  // namespace rpm

// rpm_group.hip — ONE process, N GPUs: the mesh intervals of one NLP sharded over the devices of a node behind the C ABI
// (rpm_group_*, include/rpm_hip.h).  The caller is lpopc's single-process NLPSolver::SolveNlp (Core/LpNLPSolver.cpp:13-53):
// it creates ONE TNLP object and Ipopt calls it from one thread, so a second GPU is only reachable from inside that call.
// The reference has no counterpart (one process, one thread, no device).
//
// A group is one interval-sharded engine per listed device (rank r computes a contiguous run of every phase's tiles,
// rank 0 also the endpoint rows: rpm_shard.cpp).  Three data paths, none with a reduction (results are bit-identical to a
// single engine's):
//   * host consumer (rpm_group_eval_g / _eval_jac_g / _eval_pair — what Ipopt consumes): x, g and values are the caller's
//     arrays, page-locked once for all devices (librpm_pin.so, "portable" registrations); every device reads x from them
//     and stores ITS rows of g and — by difference, option "delta_values" — its runs of `values` straight into them over
//     its own PCIe link.  All devices are started before any is waited for;
//   * device consumer on one device (rpm_group_eval_pair_dev): x, g, values live in the HBM of the `home` rank; the other
//     ranks' tile kernels read x from and store their rows / runs into them directly over xGMI (peer access) — no pack,
//     no gather, no second kernel;
//   * all-gather (rpm_group_allgather_pair_dev): every rank has full-size arrays of its own; its tile kernel fills its
//     rows / runs there and ONE push kernel stores them into the same places of every peer's arrays, one xGMI link per
//     peer (plan a: a direct one-shot all-gather instead of a ring).
// The same device may be listed more than once (tests rehearse an 8-way group on one GPU).
#include <algorithm>
#include <new>

#include "rpm_device_internal.hpp"

namespace rpm {

struct PushSeg { long long off; int len; int which; };   // which: 0 = g, 1 = values
struct PeerPtrs { double* g[RPM_GROUP_MAX]; double* v[RPM_GROUP_MAX]; int n; };

// block (s, chunk, peer * B + instance): segment s of this rank, stored into peer's arrays at the same place
__global__ __launch_bounds__(256) void rpm_peer_push_kernel(const PushSeg* __restrict__ tab, const double* __restrict__ my_g,
                                                            const double* __restrict__ my_v, const PeerPtrs peers, int B,
                                                            long long sg, long long sv) {
  const PushSeg s = tab[blockIdx.x];
  const int peer = blockIdx.z / B, b = blockIdx.z % B;
  const long long at = s.off + (long long)b * (s.which ? sv : sg);
  const double* __restrict__ src = (s.which ? my_v : my_g) + at;
  double* __restrict__ dst = (s.which ? peers.v[peer] : peers.g[peer]) + at;
  for (int i = blockIdx.y * 256 + threadIdx.x; i < s.len; i += gridDim.y * 256) dst[i] = src[i];
}

}  // namespace rpm

using rpm::Engine;

struct rpm_group {
  std::vector<rpm_engine*> eng;
  std::vector<int> dev;
  std::string err;
  std::vector<void*> push_tab;   // per rank: its PushSeg table in its device's HBM (built on first use)
  std::vector<int> push_n;
  bool peers_ready = false;
};

static thread_local std::string g_group_create_error;

static int gfail(rpm_group* g, int code, const std::string& msg) {
  g->err = msg;
  return code;
}
static int gfail_from(rpm_group* g, int rank, int code) {
  g->err = "rank " + std::to_string(rank) + " (device " + std::to_string(g->dev[rank]) + "): " + rpm_last_error(g->eng[rank]);
  return code;
}

// every device of the group may address every other's HBM (the same device listed twice needs nothing)
static int enable_peers(rpm_group* g) {
  if (g->peers_ready) return RPM_OK;
  for (size_t i = 0; i < g->dev.size(); ++i)
    for (size_t j = 0; j < g->dev.size(); ++j) {
      if (g->dev[i] == g->dev[j]) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, g->dev[i], g->dev[j]) != hipSuccess || !can)
        return gfail(g, RPM_E_DEVICE, "device " + std::to_string(g->dev[i]) + " cannot address device " + std::to_string(g->dev[j]) + " (no peer access)");
      if (hipSetDevice(g->dev[i]) != hipSuccess) return gfail(g, RPM_E_DEVICE, "hipSetDevice failed");
      const hipError_t s = hipDeviceEnablePeerAccess(g->dev[j], 0);
      if (s != hipSuccess && s != hipErrorPeerAccessAlreadyEnabled)
        return gfail(g, RPM_E_DEVICE, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(s));
      (void)hipGetLastError();
    }
  g->peers_ready = true;
  return RPM_OK;
}

extern "C" {

int rpm_group_create(const rpm_problem_desc* desc, int n_devices, const int* device_ids, rpm_group** out) {
  if (!out) return RPM_E_INVALID;
  *out = nullptr;
  if (!desc || !device_ids || n_devices < 1 || n_devices > RPM_GROUP_MAX) {
    g_group_create_error = "rpm_group_create: need 1 .. RPM_GROUP_MAX devices";
    return RPM_E_INVALID;
  }
  rpm_group* g = new (std::nothrow) rpm_group();
  if (!g) return RPM_E_INVALID;
  for (int r = 0; r < n_devices; ++r) {
    rpm_problem_desc d = *desc;
    d.shard_mode = n_devices > 1 ? RPM_SHARD_INTERVALS : d.shard_mode;
    d.shard_rank = n_devices > 1 ? r : d.shard_rank;
    d.shard_world = n_devices > 1 ? n_devices : d.shard_world;
    rpm_engine* e = nullptr;
    const int rc = rpm_create(&d, &e);
    if (rc != RPM_OK) {
      g_group_create_error = std::string("rpm_group_create, rank ") + std::to_string(r) + ": " + rpm_last_error(nullptr);
      rpm_group_destroy(g);
      return rc;
    }
    g->eng.push_back(e);
    g->dev.push_back(device_ids[r]);
    // the caller's arrays are addressed by every device: page-locked once (shared registrations), `values` by difference
    rpm_set_option(e, "pin_host", 1);
    if (n_devices > 1) rpm_set_option(e, "delta_values", 1);
  }
  g->push_tab.assign(size_t(n_devices), nullptr);
  g->push_n.assign(size_t(n_devices), 0);
  *out = g;
  return RPM_OK;
}

void rpm_group_destroy(rpm_group* g) {
  if (!g) return;
  for (size_t r = 0; r < g->eng.size(); ++r) {
    if (r < g->push_tab.size() && g->push_tab[r]) {
      (void)hipSetDevice(g->dev[r]);
      (void)hipFree(g->push_tab[r]);
    }
    rpm_destroy(g->eng[r]);
  }
  delete g;
}

const char* rpm_group_last_error(const rpm_group* g) { return g ? g->err.c_str() : g_group_create_error.c_str(); }
int rpm_group_size(const rpm_group* g) { return g ? int(g->eng.size()) : 0; }
rpm_engine* rpm_group_engine(rpm_group* g, int rank) {
  return (g && rank >= 0 && rank < int(g->eng.size())) ? g->eng[size_t(rank)] : nullptr;
}

int rpm_group_device_init(rpm_group* g) {
  if (!g) return RPM_E_INVALID;
  for (size_t r = 0; r < g->eng.size(); ++r) {
    const int rc = rpm_device_init(g->eng[r], g->dev[r]);
    if (rc) return gfail_from(g, int(r), rc);
  }
  return RPM_OK;
}

int rpm_group_set_option(rpm_group* g, const char* key, int value) {
  if (!g || !key) return RPM_E_INVALID;
  // ("pin_host" 0 is how a caller lets go of its arrays before freeing them; the next host-consumer call of a group of
  // several devices turns "pin_host" and "delta_values" on again: each rank can only store its own share in place)
  for (size_t r = 0; r < g->eng.size(); ++r) {
    const int rc = rpm_set_option(g->eng[r], key, value);
    if (rc) return gfail_from(g, int(r), rc);
  }
  return RPM_OK;
}

// ---- host consumer: the TNLP callbacks (Core/LpopcIpopt.cpp:106-217) with every device on the data path -------------------
static int ensure_devices(rpm_group* g) {
  for (size_t r = 0; r < g->eng.size(); ++r)
    if (!g->eng[r]->e.dev) {
      const int rc = rpm_device_init(g->eng[r], g->dev[r]);
      if (rc) return gfail_from(g, int(r), rc);
    }
  return RPM_OK;
}

static int group_cons(rpm_group* g, const double* x, int new_x, double* gv, double* values, const char* who) {
  int rc = ensure_devices(g);
  if (rc) return rc;
  const size_t R = g->eng.size();
  std::vector<rpm::ConsCall> call(R);
  size_t begun = 0;
  int first_rc = RPM_OK, first_rank = -1;
  for (size_t r = 0; r < R; ++r) {   // every device started ...
    Engine& e = g->eng[r]->e;
    if (R > 1 && (!e.opt_pin_host || !e.opt_delta_values)) {
      e.opt_pin_host = 1;
      e.opt_delta_values = 1;
    }
    rc = rpm::host_cons_begin(e, x, new_x, gv, values, &call[r]);
    if (rc) { first_rc = rc; first_rank = int(r); break; }
    ++begun;
  }
  for (size_t r = 0; r < begun; ++r) {   // ... before any is waited for; all that were started are waited for
    rc = rpm::host_cons_end(g->eng[r]->e, gv, values, call[r], who);
    if (rc && first_rc == RPM_OK) { first_rc = rc; first_rank = int(r); }
  }
  if (first_rc) return gfail_from(g, first_rank, first_rc);
  return RPM_OK;
}

int rpm_group_eval_g(rpm_group* g, int n, const double* x, int new_x, int m, double* gv) {
  if (!g || g->eng.empty()) return RPM_E_INVALID;
  const Engine& e0 = g->eng[0]->e;
  if (n != e0.n || m != e0.m || !x || !gv) return gfail(g, RPM_E_INVALID, "group eval_g: size mismatch or NULL pointer");
  return group_cons(g, x, new_x, gv, nullptr, "eval_g");
}

int rpm_group_eval_jac_g(rpm_group* g, int n, const double* x, int new_x, int m, int nele_jac, int* iRow, int* jCol, double* values) {
  if (!g || g->eng.empty()) return RPM_E_INVALID;
  const Engine& e0 = g->eng[0]->e;
  if (n != e0.n || m != e0.m || nele_jac != e0.nnz_jac) return gfail(g, RPM_E_INVALID, "group eval_jac_g: size mismatch");
  if (!values) {   // structure pass (LpopcIpopt.cpp:156-164): host only, the same on every rank
    const int rc = rpm_eval_jac_g(g->eng[0], n, x, new_x, m, nele_jac, iRow, jCol, nullptr);
    return rc ? gfail_from(g, 0, rc) : RPM_OK;
  }
  if (!x) return gfail(g, RPM_E_INVALID, "group eval_jac_g: x is NULL");
  return group_cons(g, x, new_x, nullptr, values, "eval_jac_g");
}

int rpm_group_eval_pair(rpm_group* g, int n, const double* x, int m, double* gv, int nele_jac, double* values) {
  if (!g || g->eng.empty()) return RPM_E_INVALID;
  const Engine& e0 = g->eng[0]->e;
  if (n != e0.n || m != e0.m || nele_jac != e0.nnz_jac || !x || !gv || !values)
    return gfail(g, RPM_E_INVALID, "group eval_pair: size mismatch or NULL pointer");
  return group_cons(g, x, 1, gv, values, "eval_pair");
}

// objective, gradient and exact Hessian are not sharded (one small kernel each): rank 0 evaluates them
int rpm_group_eval_f(rpm_group* g, int n, const double* x, int new_x, double* obj_value) {
  if (!g || g->eng.empty()) return RPM_E_INVALID;
  const int rc = rpm_eval_f(g->eng[0], n, x, new_x, obj_value);
  return rc ? gfail_from(g, 0, rc) : RPM_OK;
}
int rpm_group_eval_grad_f(rpm_group* g, int n, const double* x, int new_x, double* grad_f) {
  if (!g || g->eng.empty()) return RPM_E_INVALID;
  const int rc = rpm_eval_grad_f(g->eng[0], n, x, new_x, grad_f);
  return rc ? gfail_from(g, 0, rc) : RPM_OK;
}

int rpm_group_eval_h(rpm_group* g, int n, const double* x, int new_x, double obj_factor, int m, const double* lambda, int new_lambda,
                     int nele_hess, int* iRow, int* jCol, double* values) {
  if (!g || g->eng.empty()) return RPM_E_INVALID;
  const int rc = rpm_eval_h(g->eng[0], n, x, new_x, obj_factor, m, lambda, new_lambda, nele_hess, iRow, jCol, values);
  return rc ? gfail_from(g, 0, rc) : RPM_OK;
}

// ---- device consumer ---------------------------------------------------------------------------------------------------
static int launch_all(rpm_group* g, const double* const* d_x, double* const* d_g, double* const* d_values, bool one_home) {
  const size_t R = g->eng.size();
  for (size_t r = 0; r < R; ++r) {
    Engine& e = g->eng[r]->e;
    if (hipSetDevice(g->dev[r]) != hipSuccess) return gfail(g, RPM_E_DEVICE, "hipSetDevice failed");
    const size_t k = one_home ? 0 : r;
    const int rc = rpm::dev_eval_cons(e, d_x[k], d_g[k], d_values[k], 3 | 4, rpm::dev_stream(e));
    if (rc) return gfail_from(g, int(r), rc);
  }
  return RPM_OK;
}
static int sync_all(rpm_group* g) {
  int first = RPM_OK;
  for (size_t r = 0; r < g->eng.size(); ++r) {
    (void)hipSetDevice(g->dev[r]);
    const int rc = rpm::dev_sync(g->eng[r]->e);
    if (rc && !first) first = gfail_from(g, int(r), rc);
  }
  return first;
}

int rpm_group_eval_pair_dev(rpm_group* g, int home, const double* d_x, double* d_g, double* d_values) {
  if (!g || g->eng.empty()) return RPM_E_INVALID;
  if (home < 0 || home >= int(g->eng.size()) || !d_x || !d_g || !d_values) return gfail(g, RPM_E_INVALID, "group eval_pair_dev: bad home rank or NULL pointer");
  int rc = ensure_devices(g);
  if (!rc) rc = enable_peers(g);
  if (rc) return rc;
  rc = launch_all(g, &d_x, &d_g, &d_values, true);
  const int rs = sync_all(g);   // whatever was launched is waited for
  return rc ? rc : rs;
}

static int build_push_table(rpm_group* g, int r) {
  if (g->push_tab[size_t(r)]) return RPM_OK;
  Engine& e = g->eng[size_t(r)]->e;
  std::vector<rpm::PushSeg> tab;
  for (int which = 0; which < 2; ++which)
    for (const rpm_segment& s : rpm::shard_segments(e, which, r, nullptr)) tab.push_back(rpm::PushSeg{s.off, s.len, which});
  g->push_n[size_t(r)] = int(tab.size());
  if (hipSetDevice(g->dev[size_t(r)]) != hipSuccess) return gfail(g, RPM_E_DEVICE, "hipSetDevice failed");
  if (hipMalloc(&g->push_tab[size_t(r)], (tab.size() ? tab.size() : 1) * sizeof(rpm::PushSeg)) != hipSuccess) return gfail(g, RPM_E_DEVICE, "hipMalloc (push table)");
  if (!tab.empty() && hipMemcpy(g->push_tab[size_t(r)], tab.data(), tab.size() * sizeof(rpm::PushSeg), hipMemcpyHostToDevice) != hipSuccess)
    return gfail(g, RPM_E_DEVICE, "hipMemcpy (push table)");
  return RPM_OK;
}

int rpm_group_allgather_pair_dev(rpm_group* g, const double* const* d_x, double* const* d_g, double* const* d_values) {
  if (!g || g->eng.empty() || !d_x || !d_g || !d_values) return RPM_E_INVALID;
  const size_t R = g->eng.size();
  for (size_t r = 0; r < R; ++r)
    if (!d_x[r] || !d_g[r] || !d_values[r]) return gfail(g, RPM_E_INVALID, "group allgather_pair_dev: NULL pointer");
  int rc = ensure_devices(g);
  if (!rc) rc = enable_peers(g);
  for (size_t r = 0; r < R && !rc; ++r) rc = build_push_table(g, int(r));
  if (rc) return rc;
  rc = launch_all(g, d_x, d_g, d_values, false);   // every rank's rows / runs into its own arrays ...
  for (size_t r = 0; r < R && !rc && R > 1; ++r) {  // ... and from there into every peer's, same stream: ordered after the tile kernel
    Engine& e = g->eng[r]->e;
    if (!g->push_n[r]) continue;
    rpm::PeerPtrs peers{};
    for (size_t p = 0; p < R; ++p)
      if (p != r && d_g[p] != d_g[r]) { peers.g[peers.n] = d_g[p]; peers.v[peers.n] = d_values[p]; ++peers.n; }
    if (!peers.n) continue;
    if (hipSetDevice(g->dev[r]) != hipSuccess) { rc = gfail(g, RPM_E_DEVICE, "hipSetDevice failed"); break; }
    const int B = e.n_instances;
    hipLaunchKernelGGL(rpm::rpm_peer_push_kernel, dim3(unsigned(g->push_n[r]), 2, unsigned(peers.n * B)), dim3(256), 0,
                       static_cast<hipStream_t>(rpm::dev_stream(e)), static_cast<const rpm::PushSeg*>(g->push_tab[r]), d_g[r],
                       d_values[r], peers, B, e.stride_g(), e.stride_values());
    if (hipGetLastError() != hipSuccess) rc = gfail(g, RPM_E_DEVICE, "rpm_peer_push_kernel launch failed");
  }
  const int rs = sync_all(g);
  return rc ? rc : rs;
}

}  // extern "C"

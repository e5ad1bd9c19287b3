// rpm_sweep.cpp — rpm_sweep_*: the batched device solver (rpm_ipm_*, row f-2) over several GPUs from ONE process.  The B
// independent instances of one transcription (the MPC sweep of BASELINE config 5) are dealt to the listed devices in contiguous
// shares; every device has its own engine and solver, a call runs them side by side on a host thread each (the solver's loop
// blocks on its stream's counters).  Nothing crosses between devices: an instance's result is what a single engine computes for
// it.  No reference counterpart (lpopc hands one NLP to Ipopt, Core/LpNLPSolver.cpp:13-53).  Host only.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rpm_hip.h"

struct rpm_sweep {
  std::vector<rpm_engine*> eng;
  std::vector<rpm_ipm*> ipm;
  std::vector<int> dev, first, count;   // device, first instance and number of instances of every share
  int B = 0, n = 0, m = 0;
  std::string err;
};

static std::string g_sweep_create_error;

extern "C" {

void rpm_sweep_destroy(rpm_sweep* s) {
  if (!s) return;
  for (size_t r = 0; r < s->eng.size(); ++r) {
    if (r < s->ipm.size() && s->ipm[r]) rpm_ipm_destroy(s->ipm[r]);
    if (s->eng[r]) rpm_destroy(s->eng[r]);
  }
  delete s;
}

int rpm_sweep_create(const rpm_problem_desc* desc, int n_devices, const int* device_ids, rpm_sweep** out) {
  if (!out) return RPM_E_INVALID;
  *out = nullptr;
  if (!desc || !device_ids || n_devices < 1 || n_devices > RPM_GROUP_MAX || desc->n_instances < n_devices) {
    g_sweep_create_error = "rpm_sweep_create: need 1 .. RPM_GROUP_MAX devices and at least one instance per device";
    return RPM_E_INVALID;
  }
  rpm_sweep* s = new (std::nothrow) rpm_sweep();
  if (!s) return RPM_E_INVALID;
  s->B = desc->n_instances;
  for (int r = 0; r < n_devices; ++r) {
    const int i0 = int((long long)s->B * r / n_devices), i1 = int((long long)s->B * (r + 1) / n_devices);
    rpm_problem_desc d = *desc;
    d.n_instances = i1 - i0;
    rpm_engine* e = nullptr;
    int rc = rpm_create(&d, &e);
    if (rc == RPM_OK) {
      s->eng.push_back(e);
      rc = rpm_device_init(e, device_ids[r]);
    }
    rpm_ipm* p = nullptr;
    if (rc == RPM_OK) rc = rpm_ipm_create(e, &p);
    if (rc != RPM_OK) {
      g_sweep_create_error = std::string("rpm_sweep_create, device ") + std::to_string(device_ids[r]) + ": " + rpm_last_error(e);
      rpm_sweep_destroy(s);
      return rc;
    }
    s->ipm.push_back(p);
    s->dev.push_back(device_ids[r]);
    s->first.push_back(i0);
    s->count.push_back(i1 - i0);
  }
  int nnz_j = 0, nnz_h = 0, style = 0;
  rpm_get_nlp_info(s->eng[0], &s->n, &s->m, &nnz_j, &nnz_h, &style);
  *out = s;
  return RPM_OK;
}

const char* rpm_sweep_last_error(const rpm_sweep* s) { return s ? s->err.c_str() : g_sweep_create_error.c_str(); }
int rpm_sweep_size(const rpm_sweep* s) { return s ? int(s->eng.size()) : 0; }
rpm_engine* rpm_sweep_engine(rpm_sweep* s, int share) { return (s && share >= 0 && share < int(s->eng.size())) ? s->eng[size_t(share)] : nullptr; }
rpm_ipm* rpm_sweep_solver(rpm_sweep* s, int share) { return (s && share >= 0 && share < int(s->ipm.size())) ? s->ipm[size_t(share)] : nullptr; }
int rpm_sweep_share(const rpm_sweep* s, int share, int* first_instance, int* n_instances) {
  if (!s || share < 0 || share >= int(s->eng.size())) return RPM_E_INVALID;
  if (first_instance) *first_instance = s->first[size_t(share)];
  if (n_instances) *n_instances = s->count[size_t(share)];
  return RPM_OK;
}

static int sfail(rpm_sweep* s, int r, int rc) {
  s->err = "share " + std::to_string(r) + " (device " + std::to_string(s->dev[size_t(r)]) + "): " + rpm_ipm_last_error(s->ipm[size_t(r)]);
  return rc;
}

int rpm_sweep_set_option(rpm_sweep* s, const char* key, double value) {
  if (!s || !key) return RPM_E_INVALID;
  for (size_t r = 0; r < s->ipm.size(); ++r) {
    const int rc = rpm_ipm_set_option(s->ipm[r], key, value);
    if (rc) return sfail(s, int(r), rc);
  }
  return RPM_OK;
}

int rpm_sweep_set_bounds(rpm_sweep* s, int instance, const double* x_l, const double* x_u) {
  if (!s || instance < 0 || instance >= s->B) return RPM_E_INVALID;
  for (size_t r = 0; r < s->ipm.size(); ++r)
    if (instance < s->first[r] + s->count[r]) {
      (void)hipSetDevice(s->dev[r]);
      const int rc = rpm_ipm_set_bounds(s->ipm[r], instance - s->first[r], x_l, x_u);
      return rc ? sfail(s, int(r), rc) : RPM_OK;
    }
  return RPM_E_INVALID;
}

/* x: B x n (starting points in, solutions out), lambda: B x m or NULL; per instance, any may be NULL: objective, status,
 * iteration count, scaled KKT error — as rpm_ipm_solve, over all shares at once */
int rpm_sweep_solve(rpm_sweep* s, double* x, double* lambda, double* obj, int* status, int* iterations, double* kkt_error) {
  if (!s || !x) return RPM_E_INVALID;
  const size_t N = s->ipm.size();
  std::vector<int> rcs(N, RPM_OK);
  auto run = [&](size_t r) {
    (void)hipSetDevice(s->dev[r]);       // the current device is per host thread
    const size_t i0 = size_t(s->first[r]);
    rcs[r] = rpm_ipm_solve(s->ipm[r], x + i0 * s->n, lambda ? lambda + i0 * s->m : nullptr, obj ? obj + i0 : nullptr,
                           status ? status + i0 : nullptr, iterations ? iterations + i0 : nullptr, kkt_error ? kkt_error + i0 : nullptr);
  };
  std::vector<std::thread> th;
  for (size_t r = 1; r < N; ++r) th.emplace_back(run, r);
  run(0);
  for (std::thread& t : th) t.join();
  for (size_t r = 0; r < N; ++r)
    if (rcs[r]) return sfail(s, int(r), rcs[r]);
  return RPM_OK;
}

/* totals over the shares of the last solve: batched iterations (the largest share's count), factorisations and trial points (sums) */
int rpm_sweep_get_stats(rpm_sweep* s, int* iterations, int* factorizations, int* trial_points) {
  if (!s) return RPM_E_INVALID;
  int it = 0, fa = 0, tr = 0;
  for (rpm_ipm* p : s->ipm) {
    int a = 0, b = 0, c = 0;
    const int rc = rpm_ipm_get_stats(p, &a, &b, &c);
    if (rc) return rc;
    it = std::max(it, a); fa += b; tr += c;
  }
  if (iterations) *iterations = it;
  if (factorizations) *factorizations = fa;
  if (trial_points) *trial_points = tr;
  return RPM_OK;
}

}  // extern "C"

// rpm_host_path.hip — the host-pointer (Ipopt-facing) evaluation path behind rpm_eval_g / rpm_eval_jac_g / rpm_eval_pair:
// what LpopcIpopt::eval_g and eval_jac_g (Core/LpopcIpopt.cpp:135-181) do with their `new double[n]` copy, the
// NLPWrapper call and the element-wise copy-out, done here with as few PCIe round trips and queue operations as the
// TNLP protocol allows.  Measured floors on this box (tools/ubench/pcie_paths.hip, profiles/r02_pcie_paths.jsonl):
// one queue operation + hipStreamSynchronize 10.2 us, PCIe 57 GB/s marginal for a copy-engine D2H (47.9 GB/s at the
// NL prefix's 3.1 MB), 46 GB/s for stores a kernel makes straight into page-locked host memory.
//
//   * transport of the caller's arrays: with option "pin_host" (RpmTNLP turns it on for Ipopt's arrays) their pages are
//     registered once in the process-wide table of librpm_pin.so and the device addresses them; otherwise — the default —
//     the CPU copies x into, and g / values out of, page-locked staging buffers this engine owns.  Caller memory that is
//     not registered never reaches the HIP runtime (its pageable-copy path would lock the pages itself and cache the lock);
//   * x is read by the tile kernel straight from the page-locked array (the caller's or the staging copy), g is stored
//     straight into it (no copy-engine operations, ONE launch + ONE synchronisation per rpm_eval_g);
//   * NaN/Inf detection is fused into the kernel that produces the values (K.chk; two host-visible words);
//   * the Jacobian of the same x is produced by the same launch into HBM (fused pair) and delivered by
//     rpm_eval_jac_g(new_x = 0) either by one copy-engine transfer (default; option "const_once": NL prefix only) or,
//     with option "delta_values", by a kernel that stores only the 4-KB runs whose bits differ from what this engine
//     last delivered into the same array (the constant Doffdiag block, the linear entries and every block of the
//     finite-difference Jacobian that does not depend on x never cross PCIe again);
//   * rpm_eval_pair is both callbacks in one call: one launch for g + Jacobian, one delivery kernel, one
//     synchronisation.
// There is no CPU arithmetic here and no fallback: without a device every entry point fails with RPM_E_DEVICE.
#include "rpm_device_internal.hpp"

namespace rpm {

void host_new_x(Engine& e);

// ------------------------------------------------------------------------------------------------------------------
// Delivery of `values` by difference.  runs[r] = {off, len} (doubles, len <= DELTA_RUN) tile the part of the flat
// values array this engine owns.  One workgroup per run: compare the fresh values with the mirror of what the host
// array holds; if any bit differs (or `force`), store the run into the host array and refresh the mirror.
constexpr int DELTA_RUN = 512;
struct RunDev { long long off; int len, pad; };

__global__ __launch_bounds__(256) void rpm_delta_copy_kernel(const RunDev* __restrict__ runs, const double* __restrict__ fresh,
                                                             unsigned long long* __restrict__ mirror,
                                                             double* __restrict__ host, int force,
                                                             unsigned* __restrict__ sent_runs) {
  const RunDev r = runs[blockIdx.x];
  const int t = threadIdx.x;
  const unsigned long long* f = reinterpret_cast<const unsigned long long*>(fresh) + r.off;
  unsigned long long* mi = mirror + r.off;
  unsigned long long* h = reinterpret_cast<unsigned long long*>(host) + r.off;
  // a thread takes two neighbouring entries: 16-byte stores over PCIe (1 KB per wave instruction) where the run starts on a
  // 16-byte boundary of all three arrays, else the entries t and t + 256
  const bool wide = ((reinterpret_cast<size_t>(f) | reinterpret_cast<size_t>(mi) | reinterpret_cast<size_t>(h)) & 15) == 0;
  const int i0 = wide ? 2 * t : t, i1 = wide ? 2 * t + 1 : t + 256;
  const bool in0 = i0 < r.len, in1 = i1 < r.len;
  unsigned long long a0 = 0ull, a1 = 0ull;
  if (wide && in1) { const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(f + i0); a0 = v.x; a1 = v.y; }
  else { a0 = in0 ? f[i0] : 0ull; a1 = in1 ? f[i1] : 0ull; }
  int differ = force;
  if (!force) {
    unsigned long long m0 = 0ull, m1 = 0ull;
    if (wide && in1) { const ulonglong2 v = *reinterpret_cast<const ulonglong2*>(mi + i0); m0 = v.x; m1 = v.y; }
    else { m0 = in0 ? mi[i0] : 0ull; m1 = in1 ? mi[i1] : 0ull; }
    differ = (a0 != m0) || (a1 != m1);
  }
  if (!__syncthreads_or(differ)) return;
  if (wide && in1) {
    *reinterpret_cast<ulonglong2*>(h + i0) = make_ulonglong2(a0, a1);
    *reinterpret_cast<ulonglong2*>(mi + i0) = make_ulonglong2(a0, a1);
  } else {
    if (in0) { h[i0] = a0; mi[i0] = a0; }
    if (in1) { h[i1] = a1; mi[i1] = a1; }
  }
  if (sent_runs && t == 0) atomicAdd(sent_runs, 1u);
}

struct HostPath {
  // delta delivery
  RunDev* d_runs = nullptr;
  int n_runs = 0;
  unsigned long long* d_mirror = nullptr;
  const double* delta_host = nullptr;      // host array whose content the mirror describes
  std::vector<long long> sample_idx;       // positions (inside the owned ranges) checked before every delta delivery
  std::vector<unsigned long long> sample_val;
  unsigned* d_sent = nullptr;              // statistics: runs stored so far (device counter, read on request only)
  unsigned sent_read = 0;                  // its value at the last query
  long long owned = 0;                     // doubles this engine owns in `values`
  // objective: the value(s) land in page-locked host memory straight from the summing kernel
  double* h_obj = nullptr;
  double* d_obj_alias = nullptr;
  bool obj_valid = false;                  // h_obj and the engine's gradient buffer hold f and grad f of the current x
  // "const_once": 16 sampled entries of the linear / constant tail as this engine left them in the caller's array
  std::vector<unsigned long long> tail_val;
  // results waiting in staging slots for the call's synchronisation: then copied into the caller's arrays by the CPU
  struct CopyOut { double* dst; const double* src; size_t count; };
  std::vector<CopyOut> pending;
};

static HostPath& host_path(Engine& e) {
  Device& d = *e.dev;
  if (!d.host_path) d.host_path = new HostPath();
  return *static_cast<HostPath*>(d.host_path);
}
void host_path_destroy(Device* d) {
  if (!d->host_path) return;
  HostPath* h = static_cast<HostPath*>(d->host_path);
  if (h->d_runs) (void)hipFree(h->d_runs);
  if (h->d_mirror) (void)hipFree(h->d_mirror);
  if (h->d_sent) (void)hipFree(h->d_sent);
  if (h->h_obj) (void)hipHostFree(h->h_obj);
  delete h;
  d->host_path = nullptr;
}

// the call's one synchronisation, then the CPU half of the staged deliveries
static int sync_and_deliver(Engine& e) {
  HIP_TRY(e, hipStreamSynchronize(e.dev->stream));
  dev_stage_synced(e);
  HostPath& h = host_path(e);
  for (const HostPath::CopyOut& c : h.pending) std::memcpy(c.dst, c.src, c.count * sizeof(double));
  h.pending.clear();
  return RPM_OK;
}

// x for a kernel: the caller's registered array, or the staging copy (read in place with "zero_copy", else moved to HBM)
static int stage_x_in(Engine& e, const double* x, const double** xa) {
  Device& d = *e.dev;
  const size_t count = size_t(e.n_instances) * e.n;
  void* alias = dev_pin_host(e, x, count * sizeof(double));
  if (alias && e.opt_zero_copy) { *xa = static_cast<const double*>(alias); return RPM_OK; }
  if (alias) {
    HIP_TRY(e, hipMemcpyAsync(d.d_x, x, count * sizeof(double), hipMemcpyHostToDevice, d.stream));
    *xa = d.d_x;
    return RPM_OK;
  }
  double *hx = nullptr, *ax = nullptr;
  int rc = dev_stage_reserve(e, STAGE_X, count, &hx, &ax);
  if (rc) return rc;
  std::memcpy(hx, x, count * sizeof(double));
  d.stage[STAGE_X].busy = true;
  if (e.opt_zero_copy) { *xa = ax; return RPM_OK; }
  HIP_TRY(e, hipMemcpyAsync(d.d_x, hx, count * sizeof(double), hipMemcpyHostToDevice, d.stream));
  *xa = d.d_x;
  return RPM_OK;
}

// queue the delivery of `count` doubles at dev into the caller's array `host`: straight by the copy engine when the
// array is registered, else into the staging slot (+ a CPU copy after the synchronisation)
static int stage_out(Engine& e, int slot, double* host, const double* dev, size_t count) {
  Device& d = *e.dev;
  if (count == 0) return RPM_OK;
  if (dev_pin_host(e, host, count * sizeof(double))) {
    HIP_TRY(e, hipMemcpyAsync(host, dev, count * sizeof(double), hipMemcpyDeviceToHost, d.stream));
    return RPM_OK;
  }
  double* h = nullptr;
  int rc = dev_stage_reserve(e, slot, count, &h, nullptr);
  if (rc) return rc;
  HIP_TRY(e, hipMemcpyAsync(h, dev, count * sizeof(double), hipMemcpyDeviceToHost, d.stream));
  host_path(e).pending.push_back(HostPath::CopyOut{host, h, count});
  return RPM_OK;
}

static int ensure_flags(Engine& e) {
  Device& d = *e.dev;
  if (!d.h_flags2) {   // two words in page-locked host memory the device can write: no memset, no copy back
    HIP_TRY(e, hipHostMalloc(reinterpret_cast<void**>(&d.h_flags2), 4 * sizeof(int), hipHostMallocMapped));
    HIP_TRY(e, hipHostGetDevicePointer(reinterpret_cast<void**>(&d.d_flags2), d.h_flags2, 0));
    d.h_flags2[0] = d.h_flags2[1] = d.h_flags2[2] = d.h_flags2[3] = 0;   // words: g, Jacobian, (unused), objective gradient
  }
  return RPM_OK;
}

static int delta_setup(Engine& e) {
  HostPath& h = host_path(e);
  if (h.d_runs) return RPM_OK;
  const long long nnz = e.nnz_jac, B = e.n_instances;
  std::vector<std::pair<long long, long long>> ranges;   // owned ranges of one instance
  if (e.shard_mode == RPM_SHARD_INTERVALS && e.shard_world > 1) {
    for (const rpm_segment& s : shard_segments(e, 1, e.shard_rank, nullptr)) ranges.emplace_back(s.off, s.len);
  } else {
    ranges.emplace_back(0, nnz);
  }
  std::vector<RunDev> runs;
  h.owned = 0;
  for (long long b = 0; b < B; ++b)
    for (auto& rg : ranges)
      for (long long o = 0; o < rg.second; o += DELTA_RUN) {
        const int len = int(std::min<long long>(DELTA_RUN, rg.second - o));
        runs.push_back(RunDev{b * nnz + rg.first + o, len, 0});
        h.owned += len;
      }
  h.n_runs = int(runs.size());
  HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&h.d_runs), (runs.size() ? runs.size() : 1) * sizeof(RunDev)));
  if (!runs.empty()) HIP_TRY(e, hipMemcpy(h.d_runs, runs.data(), runs.size() * sizeof(RunDev), hipMemcpyHostToDevice));
  HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&h.d_mirror), size_t(B) * nnz * sizeof(double)));
  // the counter lives in HBM: an atomic per stored run into HOST memory is a PCIe round trip each (337 of them made a
  // delivery take 340 us instead of 40)
  HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(&h.d_sent), sizeof(unsigned)));
  // on the engine's stream: a null-stream hipMemset is not ordered against a non-blocking stream, and returned before
  // the fill ran -- it then zeroed the counter in the middle of the first delivery (seen once: 1618 of 1665 runs)
  HIP_TRY(e, hipMemsetAsync(h.d_sent, 0, sizeof(unsigned), e.dev->stream));
  // 64 sample positions spread over the owned runs (first and last element included)
  const int NS = 64;
  h.sample_idx.clear();
  if (!runs.empty())
    for (int k = 0; k < NS; ++k) {
      const RunDev& r = runs[size_t((long long)k * (runs.size() - 1) / (NS - 1))];
      h.sample_idx.push_back(r.off + ((k & 1) ? r.len - 1 : 0));
    }
  h.sample_val.assign(h.sample_idx.size(), 0ull);
  return RPM_OK;
}

// deliver the Jacobian values in the engine's staging buffer into the caller's array `values` (host pointer), `va` its
// device alias.  Queues one kernel on the engine's stream; the caller synchronises and then calls delta_commit.
static int delta_enqueue(Engine& e, const double* values, double* va) {
  int rc = delta_setup(e);
  if (rc) return rc;
  HostPath& h = host_path(e);
  Device& d = *e.dev;
  int force = 1;
  if (h.delta_host == values) {   // same array as last time: has the caller left it alone?
    force = 0;
    const unsigned long long* hv = reinterpret_cast<const unsigned long long*>(values);
    for (size_t k = 0; k < h.sample_idx.size(); ++k)
      if (hv[h.sample_idx[k]] != h.sample_val[k]) { force = 1; break; }
  }
  h.delta_host = nullptr;   // not trusted again until this delivery has completed (delta_commit)
  if (h.n_runs)
    hipLaunchKernelGGL(rpm_delta_copy_kernel, dim3(unsigned(h.n_runs)), dim3(256), 0, d.stream, h.d_runs, d.d_values, h.d_mirror, va,
                       force, h.d_sent);
  HIP_TRY(e, hipGetLastError());
  return RPM_OK;
}
static void delta_commit(Engine& e, const double* values) {
  HostPath& h = host_path(e);
  const unsigned long long* hv = reinterpret_cast<const unsigned long long*>(values);
  for (size_t k = 0; k < h.sample_idx.size(); ++k) h.sample_val[k] = hv[h.sample_idx[k]];
  h.delta_host = values;
  e.last_delta_total = h.n_runs;
}
// runs stored since the previous query (rpm_get_option "delta_sent_runs"); blocking, for tests and reports only
int host_delta_sent_runs(Engine& e, int* sent) {
  *sent = 0;
  if (!e.dev || !e.dev->host_path) return RPM_OK;
  HostPath& h = host_path(e);
  if (!h.d_sent) return RPM_OK;
  unsigned now = 0;
  HIP_TRY(e, hipStreamSynchronize(e.dev->stream));
  HIP_TRY(e, hipMemcpy(&now, h.d_sent, sizeof(unsigned), hipMemcpyDeviceToHost));
  *sent = int(now - h.sent_read);
  h.sent_read = now;
  return RPM_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// One launch for the constraint side of an iterate: x from the caller's array, g into the caller's array, the
// Jacobian (when `with_jac`) into the staging buffer.  Returns with everything queued; *g_pending tells the caller
// that g still has to be copied out of the staging buffer (no page-locked alias), *scan tells it that the NaN/Inf
// verdicts need the separate scan kernel (the launch was not the one-role kernel).
static int enqueue_cons(Engine& e, const double* x, double* g, bool with_g, bool with_jac) {
  Device& d = *e.dev;
  const size_t B = size_t(e.n_instances);
  HIP_TRY(e, hipSetDevice(d.device_id));
  host_path(e).pending.clear();   // a call that failed half-way leaves nothing behind
  const bool zc = e.opt_zero_copy != 0;
  const bool sharded = e.shard_mode == RPM_SHARD_INTERVALS && e.shard_world > 1;
  if (with_g && sharded && !dev_pin_host(e, g, B * e.m * sizeof(double))) {
    e.err = "host-pointer delivery of an interval-sharded engine needs option pin_host = 1 and a registrable g array (each rank stores only its own rows)";
    if (!e.pin_note.empty()) e.err += "; " + e.pin_note;
    return RPM_E_UNSUPPORTED;
  }
  const double* xa = nullptr;
  int rc = stage_x_in(e, x, &xa);
  if (rc) return rc;
  const bool fused_chk = e.opt_check_finite && dev_cons_is_one_role(e);
  double* ga = nullptr;            // where the kernel stores g when it can store it in page-locked host memory
  if (with_g && zc && (fused_chk || !e.opt_check_finite)) {
    ga = static_cast<double*>(dev_pin_host(e, g, B * e.m * sizeof(double)));
    if (!ga) {                     // not registered: the kernel stores into the staging slot, the CPU copies it out
      double* hg = nullptr;
      rc = dev_stage_reserve(e, STAGE_G, B * e.m, &hg, &ga);
      if (rc) return rc;
      host_path(e).pending.push_back(HostPath::CopyOut{g, hg, B * e.m});
    }
  }
  if (e.opt_check_finite) {
    rc = ensure_flags(e);
    if (rc) return rc;
    d.h_flags2[0] = d.h_flags2[1] = 0;   // the previous verdicts were read after their synchronisation
  }
  const int flags = (with_g ? 1 : 0) | (with_jac ? 2 : 0) | (fused_chk ? 8 : 0);
  rc = dev_eval_cons(e, xa, ga ? ga : d.d_g, d.d_values, flags, d.stream);
  if (rc) return rc;
  if (with_g && !ga) {
    rc = stage_out(e, STAGE_G, g, d.d_g, B * e.m);
    if (rc) return rc;
  }
  if (e.opt_check_finite && !fused_chk) {
    rc = dev_nonfinite_enqueue(e, d.d_g, with_g ? B * e.m : 0, d.d_values, with_jac ? B * size_t(e.nnz_jac) : 0);
    if (rc) return rc;
  }
  return RPM_OK;
}

static int ensure_device(Engine& e) {
  if (e.dev) return RPM_OK;
  return device_init(e, 0);
}

// Every registration a call needs is made BEFORE its first launch: making one may replace or let go of another of this
// engine's registrations (arrays sharing a page are merged, the least recently used goes at the cap), which must not
// happen under a kernel that is storing through it.  Every entry point ends with a synchronisation, so nothing of this
// engine is in flight here; the later dev_pin_host calls of the same call only look the aliases up.
static void acquire_registrations(Engine& e, const double* x, const double* g, const double* values, const double* grad) {
  if (!e.opt_pin_host) return;
  const size_t B = size_t(e.n_instances);
  if (x) (void)dev_pin_host(e, x, B * e.n * sizeof(double));
  if (g) (void)dev_pin_host(e, g, B * e.m * sizeof(double));
  if (values) (void)dev_pin_host(e, values, B * e.nnz_jac * sizeof(double));
  if (grad) (void)dev_pin_host(e, grad, B * e.n * sizeof(double));
}

// "const_once" trusts the tail of an array only if it is the array this engine filled last AND 16 sampled tail entries
// still hold what it stored there (an array freed and re-allocated at the same address does not pass)
static size_t tail_pos(const Engine& e, int k) {
  const size_t tail = size_t(e.nnz_jac) - size_t(e.nnz_nl);
  return size_t(e.nnz_nl) + (tail > 1 ? size_t(k) * (tail - 1) / 15 : 0);
}
static bool tail_untouched(Engine& e, const double* values) {
  HostPath& h = host_path(e);
  if (e.nnz_jac <= e.nnz_nl || h.tail_val.size() != 16) return false;
  const unsigned long long* hv = reinterpret_cast<const unsigned long long*>(values);
  for (int k = 0; k < 16; ++k)
    if (hv[tail_pos(e, k)] != h.tail_val[k]) return false;
  return true;
}
static void tail_remember(Engine& e, const double* values) {
  HostPath& h = host_path(e);
  h.tail_val.clear();
  if (!e.opt_const_once || e.n_instances != 1 || e.nnz_jac <= e.nnz_nl) return;
  const unsigned long long* hv = reinterpret_cast<const unsigned long long*>(values);
  for (int k = 0; k < 16; ++k) h.tail_val.push_back(hv[tail_pos(e, k)]);
}

// queue the delivery of the staged Jacobian values into `values`; *delta = the delivery needs delta_commit afterwards
static int enqueue_values(Engine& e, double* values, bool* delta) {
  Device& d = *e.dev;
  const size_t B = size_t(e.n_instances);
  void* va = dev_pin_host(e, values, B * e.nnz_jac * sizeof(double));
  *delta = false;
  if (e.opt_delta_values && va) {
    *delta = true;
    return delta_enqueue(e, values, static_cast<double*>(va));
  }
  if (e.shard_mode == RPM_SHARD_INTERVALS && e.shard_world > 1) {
    e.err = "host-pointer Jacobian delivery of an interval-sharded engine needs option delta_values = 1 (each rank stores only its own runs)";
    return RPM_E_UNSUPPORTED;
  }
  // values = [NL | LIN | CONST]: with "const_once" the constant tail (54 % of the entries at the metric problem) crosses
  // PCIe only when the caller hands a buffer this engine did not fill last time
  size_t count = B * e.nnz_jac;
  if (e.opt_const_once && e.n_instances == 1 && values == e.const_filled && tail_untouched(e, values)) count = size_t(e.nnz_nl);
  return stage_out(e, STAGE_V, values, d.d_values, count);
}

// ---- the constraint callbacks in two halves: everything queued / synchronised, delivered and judged.  One engine calls
// them back to back (host_eval_g / host_eval_jac_values / host_eval_pair below); a group of interval-sharded engines, one per
// GPU (rpm_group.hip), begins on every device before it ends on any, so that the devices work at the same time.
//   g != nullptr: the constraint vector is wanted; values != nullptr: the Jacobian values are wanted.
int host_cons_begin(Engine& e, const double* x, int new_x, double* g, double* values, ConsCall* c) {
  int rc = ensure_device(e);
  if (rc) return rc;
  Device& d = *e.dev;
  HIP_TRY(e, hipSetDevice(d.device_id));
  if (new_x) host_new_x(e);
  c->want_g = g != nullptr;
  c->want_v = values != nullptr;
  c->cached = !c->want_g && !new_x && d.cache_valid;   // eval_jac_g(new_x = false) after an eval_g of the same x
  c->delta = false;
  c->launched_jac = false;
  host_path(e).pending.clear();
  acquire_registrations(e, c->cached ? nullptr : x, g, values, nullptr);
  if (!c->cached) {
    d.cache_valid = false;
    c->launched_jac = c->want_v || e.opt_fuse_pair != 0;   // the Jacobian of the same x comes out of the same launch
    rc = enqueue_cons(e, x, g, c->want_g, c->launched_jac);
    if (rc) return rc;
    e.jac_nonfinite = -1;
  }
  if (c->want_v) {
    rc = enqueue_values(e, values, &c->delta);
    if (rc) return rc;
  }
  return RPM_OK;
}

int host_cons_end(Engine& e, double* g, double* values, const ConsCall& c, const char* who) {
  Device& d = *e.dev;
  HIP_TRY(e, hipSetDevice(d.device_id));
  int rc = sync_and_deliver(e);   // the call's one synchronisation
  if (rc) return rc;
  if (c.want_v) {
    if (c.delta) delta_commit(e, values);
    e.const_filled = values;
    tail_remember(e, values);
  }
  if (!c.cached) {
    d.cache_valid = c.launched_jac;
    if (e.opt_check_finite && c.launched_jac) e.jac_nonfinite = d.h_flags2[1];
  }
  if (!e.opt_check_finite) return RPM_OK;
  if (c.want_g && d.h_flags2[0] != 0) {
    e.err = std::string(who) + ": non-finite constraint value";
    return RPM_E_NONFINITE;
  }
  if (c.want_v) {
    if (c.cached && e.jac_nonfinite < 0) {   // the cached pair was produced with "check_finite" off: scan it now
      e.jac_nonfinite = dev_nonfinite(e, d.d_values, size_t(e.n_instances) * e.nnz_jac);
      if (e.jac_nonfinite < 0) { e.err = std::string(who) + ": the NaN/Inf scan failed"; return RPM_E_DEVICE; }
    }
    if ((c.cached ? e.jac_nonfinite : d.h_flags2[1]) != 0) {
      e.err = std::string(who) + ": non-finite Jacobian value";
      return RPM_E_NONFINITE;
    }
  }
  return RPM_OK;
}

int host_eval_g(Engine& e, const double* x, int new_x, double* g) {
  ConsCall c;
  int rc = host_cons_begin(e, x, new_x, g, nullptr, &c);
  return rc ? rc : host_cons_end(e, g, nullptr, c, "eval_g");
}
int host_eval_jac_values(Engine& e, const double* x, int new_x, double* values) {
  ConsCall c;
  int rc = host_cons_begin(e, x, new_x, nullptr, values, &c);
  return rc ? rc : host_cons_end(e, nullptr, values, c, "eval_jac_g");
}
int host_eval_pair(Engine& e, const double* x, double* g, double* values) {
  ConsCall c;
  int rc = host_cons_begin(e, x, 1, g, values, &c);
  return rc ? rc : host_cons_end(e, g, values, c, "eval_pair");
}

// ---- objective and gradient (LpopcIpopt::eval_f / eval_grad_f, Core/LpopcIpopt.cpp:106-133).  The first of the two calls
// at a new x launches ONE objective kernel that produces both (the gradient kernel computes the quadrature anyway); the
// value comes home through page-locked memory with that launch's synchronisation, the gradient stays in HBM until
// eval_grad_f asks for it (one copy).  NaN/Inf in the gradient is noted by the kernel that stores it.
void host_new_x(Engine& e) {   // any callback that receives new_x = true invalidates what the others cached
  if (!e.dev) return;
  e.dev->cache_valid = false;
  if (e.dev->host_path) host_path(e).obj_valid = false;
}

static int enqueue_obj(Engine& e, const double* x) {
  Device& d = *e.dev;
  HostPath& h = host_path(e);
  const size_t B = size_t(e.n_instances);
  HIP_TRY(e, hipSetDevice(d.device_id));
  if (!h.h_obj) {
    HIP_TRY(e, hipHostMalloc(reinterpret_cast<void**>(&h.h_obj), B * sizeof(double), hipHostMallocMapped));
    HIP_TRY(e, hipHostGetDevicePointer(reinterpret_cast<void**>(&h.d_obj_alias), h.h_obj, 0));
  }
  host_path(e).pending.clear();
  const double* xa = nullptr;
  int rc = stage_x_in(e, x, &xa);
  if (rc) return rc;
  rc = ensure_flags(e);
  if (rc) return rc;
  d.h_flags2[3] = 0;
  return dev_eval_obj(e, xa, h.d_obj_alias, d.d_grad, d.stream, e.opt_check_finite != 0);
}

int host_eval_f(Engine& e, const double* x, int new_x, double* obj) {
  int rc = ensure_device(e);
  if (rc) return rc;
  if (new_x) host_new_x(e);
  HostPath& h = host_path(e);
  if (!h.obj_valid) {
    acquire_registrations(e, x, nullptr, nullptr, nullptr);
    rc = enqueue_obj(e, x);
    if (rc) return rc;
    rc = sync_and_deliver(e);
    if (rc) return rc;
    h.obj_valid = true;
  }
  for (int b = 0; b < e.n_instances; ++b) obj[b] = h.h_obj[b];
  if (e.opt_check_finite)
    for (int b = 0; b < e.n_instances; ++b)
      if (!std::isfinite(obj[b])) { e.err = "eval_f: non-finite objective"; return RPM_E_NONFINITE; }
  return RPM_OK;
}

int host_eval_grad_f(Engine& e, const double* x, int new_x, double* grad) {
  int rc = ensure_device(e);
  if (rc) return rc;
  if (new_x) host_new_x(e);
  HostPath& h = host_path(e);
  Device& d = *e.dev;
  acquire_registrations(e, h.obj_valid ? nullptr : x, nullptr, nullptr, grad);
  if (!h.obj_valid) {
    rc = enqueue_obj(e, x);
    if (rc) return rc;
  }
  const size_t count = size_t(e.n_instances) * e.n;
  rc = stage_out(e, STAGE_GRAD, grad, d.d_grad, count);
  if (rc) return rc;
  rc = sync_and_deliver(e);
  if (rc) return rc;
  h.obj_valid = true;
  if (e.opt_check_finite && d.h_flags2[3] != 0) { e.err = "eval_grad_f: non-finite gradient"; return RPM_E_NONFINITE; }
  return RPM_OK;
}

}  // namespace rpm

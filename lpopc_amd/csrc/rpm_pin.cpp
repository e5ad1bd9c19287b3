// rpm_pin.cpp — librpm_pin.so: the ONE table of page-locked caller memory of this process.
//
// hipHostRegister is a process-wide facility with one table per process, keyed by start address.  What the runtime does
// with it was probed on the MI355X box (tools/ubench/host_register_probe.hip, profiles/r03_host_register_probe.log):
//   * it accepts ANY registration — the same range twice (the second entry replaces the first, whose pin then leaks and
//     whose second hipHostUnregister fails), ranges inside or across registered ranges, ranges sharing a page;
//   * a hipMemcpy whose host pointer lies inside a registered range is served through that registration: one that starts
//     inside and ends beyond the range fails with "invalid argument" — so a registration must cover exactly the caller's
//     array and not the rest of its first and last page, where unrelated heap objects live;
//   * a registration left behind on memory that has since been unmapped (a freed numpy array) is still honoured: a later
//     copy from whatever is mapped there now goes through the dead mapping and ends the process with "Memory access fault
//     by GPU" + abort() — the SIGABRT of round 2.
// Hence one registry for every engine of every library built from these sources (librpm_hip.so and the libraries
// lpopc_amd/userproblem.py builds around a user's functor link against this one small library; the dynamic loader maps a
// DT_NEEDED library once), which never registers a byte twice and lets go of a range exactly when nobody addresses it.
//
// Model.  An engine ("owner") holds ARRAYS: the byte ranges of the caller arrays it addresses, at most `max_holds` of
// them (least recently used goes first).  What is registered with the runtime are REGIONS: pairwise disjoint byte ranges,
// each the union of the arrays that overlap (views of one buffer, an array re-allocated at an overlapping address).
//   * a request inside a live region joins it (no second hipHostRegister, whoever made the region);
//   * a region that holds arrays of ONE owner is rebuilt from that owner's arrays whenever they change — a new array that
//     overlaps it extends it, an evicted one shrinks it — so an owner never pins more than its arrays;
//   * a region that holds arrays of several owners is frozen: nobody re-registers memory somebody else may be addressing; a
//     request that partly overlaps it is refused (the caller takes its staged path);
//   * a region is unregistered when its last array goes: rpm_destroy of one engine never unpins what another addresses;
//   * no failure is silent: every refused registration / unregistration / overlap is counted (rpm_get_option "pin_*") and its
//     text kept for rpm_last_error;
//   * arrays below RPM_PIN_MIN_BYTES are not registered at all (a staged copy of a few pages costs less than a table entry).
// An owner changes its arrays only at the start of an entry point, when nothing of it is in flight (rpm_host_path.hip,
// acquire_registrations).  What the registry cannot know is that the caller freed an array it still holds: that is the
// lifetime contract of option "pin_host" (rpm_hip.h), which is why the option is off unless the caller asks.
// Reference behaviour being replaced: LpopcIpopt's heap copy of x and element-wise copy-out of g / values
// (Core/LpopcIpopt.cpp:135-181) — no page-locking there; this is transport, no arithmetic.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "rpm_pin.h"

namespace {

struct Arr { const void* owner; uintptr_t lo, hi; unsigned long long stamp; };
struct Reg { uintptr_t lo, hi; char* dbase; bool was_shared; };   // was_shared: may be larger than its present arrays until its owner next changes them

struct Registry {
  std::mutex mu;
  std::vector<Arr> arrs;
  std::vector<Reg> regs;     // pairwise disjoint; every array lies inside exactly one
  unsigned long long clock = 0;
  long counters[RPM_PIN_N_COUNTERS] = {0};
  std::string last_error;
};

Registry& reg() {
  static Registry* r = new Registry();   // never destroyed: engines may be torn down during process exit
  return *r;
}

void note(Registry& r, int counter, const char* what, uintptr_t lo, uintptr_t hi, hipError_t s) {
  ++r.counters[counter];
  char buf[256];
  std::snprintf(buf, sizeof buf, "%s [%#llx, %#llx): %s", what, (unsigned long long)lo, (unsigned long long)hi,
                s == hipSuccess ? "refused by the registry" : hipGetErrorString(s));
  r.last_error = buf;
}

void unregister_region(Registry& r, const Reg& g) {
  const hipError_t s = hipHostUnregister(reinterpret_cast<void*>(g.lo));
  if (s != hipSuccess) {
    (void)hipGetLastError();
    note(r, RPM_PIN_UNREGISTER_FAILURES, "hipHostUnregister", g.lo, g.hi, s);
  } else {
    ++r.counters[RPM_PIN_UNREGISTERED];
  }
}

bool register_region(Registry& r, uintptr_t lo, uintptr_t hi, char** dbase) {
  // Portable: every device of the process may address it (one engine per GPU stores its runs of g / values into the
  // caller's arrays, rpm_group_*); Mapped: the kernels read x from and store g into it
  hipError_t s = hipHostRegister(reinterpret_cast<void*>(lo), size_t(hi - lo), hipHostRegisterMapped | hipHostRegisterPortable);
  if (s != hipSuccess) {
    (void)hipGetLastError();
    note(r, RPM_PIN_REGISTER_FAILURES, "hipHostRegister", lo, hi, s);
    return false;
  }
  void* d = nullptr;
  s = hipHostGetDevicePointer(&d, reinterpret_cast<void*>(lo), 0);
  if (s != hipSuccess || !d) {
    (void)hipGetLastError();
    note(r, RPM_PIN_REGISTER_FAILURES, "hipHostGetDevicePointer", lo, hi, s);
    if (hipHostUnregister(reinterpret_cast<void*>(lo)) != hipSuccess) {
      (void)hipGetLastError();
      ++r.counters[RPM_PIN_UNREGISTER_FAILURES];
    }
    return false;
  }
  ++r.counters[RPM_PIN_REGISTERED];
  *dbase = static_cast<char*>(d);
  return true;
}

const Reg* region_of(const Registry& r, uintptr_t lo, uintptr_t hi) {
  for (const Reg& g : r.regs)
    if (g.lo <= lo && hi <= g.hi) return &g;
  return nullptr;
}

bool foreign_inside(const Registry& r, const Reg& g, const void* owner) {
  for (const Arr& a : r.arrs)
    if (a.owner != owner && g.lo <= a.lo && a.hi <= g.hi) return true;
  return false;
}

// Make the regions that hold only `owner`'s arrays equal to the connected byte ranges of those arrays.  Regions that stay
// as they are are not touched; of the others the old ones are unregistered first, then the new ones registered.  Returns
// false when a registration was refused; the arrays left without a region are then removed (`must_keep`, the array being
// added, is reported through the return value).
bool reconcile(Registry& r, const void* owner) {
  // the owner's arrays outside frozen regions, as connected byte ranges
  std::vector<std::pair<uintptr_t, uintptr_t>> want;
  {
    std::vector<std::pair<uintptr_t, uintptr_t>> mine;
    for (const Arr& a : r.arrs)
      if (a.owner == owner) {
        const Reg* g = region_of(r, a.lo, a.hi);
        if (g && foreign_inside(r, *g, owner)) continue;   // lives in a frozen region
        mine.emplace_back(a.lo, a.hi);
      }
    std::sort(mine.begin(), mine.end());
    for (const auto& m : mine) {
      if (!want.empty() && m.first < want.back().second) want.back().second = std::max(want.back().second, m.second);
      else want.push_back(m);
    }
  }
  // the owner's present single-owner regions (and regions nobody holds any more)
  std::vector<size_t> drop;
  for (size_t i = 0; i < r.regs.size(); ++i) {
    const Reg& g = r.regs[i];
    if (foreign_inside(r, g, owner)) continue;
    bool keep = false;
    for (auto& w : want)
      if (w.first == g.lo && w.second == g.hi) { keep = true; w.first = w.second = 0; break; }   // exists already
    bool anyone = false;
    for (const Arr& a : r.arrs)
      if (g.lo <= a.lo && a.hi <= g.hi) { anyone = true; break; }
    // a region without arrays is nobody's: it goes.  One with this owner's arrays that is no longer wanted as it is goes too.
    if (!keep || !anyone) drop.push_back(i);
  }
  for (size_t k = drop.size(); k-- > 0;) {
    unregister_region(r, r.regs[drop[k]]);
    r.regs.erase(r.regs.begin() + drop[k]);
  }
  bool ok = true;
  for (const auto& w : want) {
    if (w.first == w.second) continue;
    char* dbase = nullptr;
    if (register_region(r, w.first, w.second, &dbase)) {
      r.regs.push_back(Reg{w.first, w.second, dbase, false});
    } else {
      ok = false;   // the arrays of this range have no region: they go
      for (size_t i = 0; i < r.arrs.size();)
        if (r.arrs[i].owner == owner && w.first <= r.arrs[i].lo && r.arrs[i].hi <= w.second) r.arrs.erase(r.arrs.begin() + i);
        else ++i;
    }
  }
  return ok;
}

int count_of(const Registry& r, const void* owner) {
  int n = 0;
  for (const Arr& a : r.arrs) n += a.owner == owner;
  return n;
}

}  // namespace

extern "C" {

void* rpm_pin_acquire(const void* owner, const void* ptr, size_t bytes, int max_holds, int also_small) {
  if (!owner || !ptr || bytes == 0 || (bytes < RPM_PIN_MIN_BYTES && !also_small)) return nullptr;
  const uintptr_t p = reinterpret_cast<uintptr_t>(ptr);
  const uintptr_t lo = p, hi = p + bytes;   // exactly the array: its first and last page also hold other people's objects
  Registry& r = reg();
  std::lock_guard<std::mutex> lock(r.mu);
  const unsigned long long now = ++r.clock;
  // an array this owner holds already (the common case: Ipopt hands the same arrays every iteration)
  for (Arr& a : r.arrs)
    if (a.owner == owner && a.lo <= lo && hi <= a.hi) {
      a.stamp = now;
      const Reg* g = region_of(r, a.lo, a.hi);
      return g ? g->dbase + (p - g->lo) : nullptr;
    }
  // make room first: the least recently used arrays of this owner go (their pages with them, below)
  bool changed = false;
  while (max_holds > 0 && count_of(r, owner) >= max_holds) {
    size_t oldest = size_t(-1);
    for (size_t i = 0; i < r.arrs.size(); ++i)
      if (r.arrs[i].owner == owner && (oldest == size_t(-1) || r.arrs[i].stamp < r.arrs[oldest].stamp)) oldest = i;
    r.arrs.erase(r.arrs.begin() + oldest);
    ++r.counters[RPM_PIN_EVICTED];
    changed = true;
  }
  // inside a live region: join it
  if (const Reg* g = region_of(r, lo, hi)) {
    if (foreign_inside(r, *g, owner)) {
      ++r.counters[RPM_PIN_SHARED];
      const_cast<Reg*>(g)->was_shared = true;
    }
    r.arrs.push_back(Arr{owner, lo, hi, now});
    char* alias = g->dbase + (p - g->lo);
    if (changed) {
      (void)reconcile(r, owner);
      const Reg* q = region_of(r, lo, hi);   // a region of this owner alone may have been rebuilt
      alias = q ? q->dbase + (p - q->lo) : nullptr;
    }
    return alias;
  }
  // partly overlapping regions: memory somebody else addresses is never re-registered under them
  bool touches = false;
  for (const Reg& g : r.regs)
    if (g.lo < hi && lo < g.hi) {
      touches = true;
      if (foreign_inside(r, g, owner)) {
        note(r, RPM_PIN_OVERLAP_REFUSED, "registration overlaps memory another engine holds", lo, hi, hipSuccess);
        if (changed) (void)reconcile(r, owner);
        return nullptr;
      }
    }
  if (touches) ++r.counters[RPM_PIN_MERGED];
  r.arrs.push_back(Arr{owner, lo, hi, now});
  if (!reconcile(r, owner)) {
    // the union with this owner's older arrays was refused (one of them may have been freed by the caller): those that
    // touch the new array were dropped with it; try the new array's pages alone
    bool present = false;
    for (const Arr& a : r.arrs) present |= (a.owner == owner && a.lo == lo && a.hi == hi);
    if (!present && touches) {
      r.arrs.push_back(Arr{owner, lo, hi, now});
      (void)reconcile(r, owner);
    }
  }
  const Reg* g = region_of(r, lo, hi);
  bool present = false;
  for (const Arr& a : r.arrs) present |= (a.owner == owner && a.lo == lo && a.hi == hi);
  return (g && present) ? g->dbase + (p - g->lo) : nullptr;
}

void rpm_pin_release_owner(const void* owner) {
  if (!owner) return;
  Registry& r = reg();
  std::lock_guard<std::mutex> lock(r.mu);
  bool any = false;
  for (size_t i = 0; i < r.arrs.size();)
    if (r.arrs[i].owner == owner) { r.arrs.erase(r.arrs.begin() + i); any = true; }
    else ++i;
  if (any) (void)reconcile(r, owner);   // regions left without arrays are unregistered; shared ones stay for their other holders
}

long rpm_pin_counter(int which) {
  Registry& r = reg();
  std::lock_guard<std::mutex> lock(r.mu);
  if (which == RPM_PIN_LIVE) return long(r.regs.size());
  if (which == RPM_PIN_LIVE_BYTES) {
    long b = 0;
    for (const Reg& g : r.regs) b += long(g.hi - g.lo);
    return b;
  }
  if (which < 0 || which >= RPM_PIN_N_COUNTERS) return -1;
  return r.counters[which];
}

int rpm_pin_held(const void* owner) {
  Registry& r = reg();
  std::lock_guard<std::mutex> lock(r.mu);
  return count_of(r, owner);
}

size_t rpm_pin_last_error(char* buf, size_t cap) {
  Registry& r = reg();
  std::lock_guard<std::mutex> lock(r.mu);
  if (buf && cap) {
    std::strncpy(buf, r.last_error.c_str(), cap - 1);
    buf[cap - 1] = 0;
  }
  return r.last_error.size();
}

}  // extern "C"

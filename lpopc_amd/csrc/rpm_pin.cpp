// rpm_pin.cpp — librpm_pin.so: the ONE table of page-locked caller memory of this process.
//
// hipHostRegister is a process-wide, page-granular facility: the runtime (and the kernel driver under it) pins whole pages
// and keeps one table per process.  Every engine of every library built from these sources — librpm_hip.so and the
// libraries lpopc_amd/userproblem.py builds around a user's functor — links against this one small library, so there is
// exactly one registry per process whatever the number of engines and libraries (the dynamic loader maps a DT_NEEDED
// library once).  What it guarantees:
//   * registrations are page-aligned supersets of the arrays they cover and never overlap each other: a request that
//     is covered by a live registration shares it (reference count per holder), a request that partly overlaps live
//     registrations held only by the requester replaces them by their union, one that partly overlaps somebody else's is
//     refused (the caller then takes the staged path);
//   * a registration is released (hipHostUnregister) exactly when its last holder lets go of it: rpm_destroy of one engine
//     never unpins pages another engine still addresses;
//   * no failure is silent: every refused registration / unregistration / overlap is counted (rpm_get_option "pin_*") and
//     its text kept for rpm_last_error;
//   * arrays below RPM_PIN_MIN_BYTES are not registered at all (a staged copy of a few pages costs less than the table entry).
// Reference behaviour being replaced: LpopcIpopt's heap copy of x and element-wise copy-out of g / values
// (Core/LpopcIpopt.cpp:135-181) — no page-locking there; this is transport, no arithmetic.
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "rpm_pin.h"

namespace {

struct Hold { const void* owner; unsigned long long stamp; };
struct Reg {
  uintptr_t lo, hi;          // page-aligned [lo, hi)
  char* dbase;               // device-visible address of lo
  std::vector<Hold> holds;   // who addresses it (one entry per owner)
};

struct Registry {
  std::mutex mu;
  std::vector<Reg> regs;     // pairwise disjoint
  unsigned long long clock = 0;
  long counters[RPM_PIN_N_COUNTERS] = {0};
  std::string last_error;
};

Registry& reg() {
  static Registry* r = new Registry();   // never destroyed: engines may be torn down during process exit
  return *r;
}

uintptr_t page_size() {
  static const uintptr_t p = uintptr_t(sysconf(_SC_PAGESIZE) > 0 ? sysconf(_SC_PAGESIZE) : 4096);
  return p;
}

void note(Registry& r, int counter, const char* what, uintptr_t lo, uintptr_t hi, hipError_t s) {
  ++r.counters[counter];
  char buf[256];
  std::snprintf(buf, sizeof buf, "%s [%#llx, %#llx): %s", what, (unsigned long long)lo, (unsigned long long)hi,
                s == hipSuccess ? "refused by the registry" : hipGetErrorString(s));
  r.last_error = buf;
}

// hipHostUnregister of one table entry (the caller removes it from the table)
void unregister_locked(Registry& r, const Reg& g) {
  const hipError_t s = hipHostUnregister(reinterpret_cast<void*>(g.lo));
  if (s != hipSuccess) {
    (void)hipGetLastError();
    note(r, RPM_PIN_UNREGISTER_FAILURES, "hipHostUnregister", g.lo, g.hi, s);
  } else {
    ++r.counters[RPM_PIN_UNREGISTERED];
  }
}

bool register_locked(Registry& r, uintptr_t lo, uintptr_t hi, char** dbase) {
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

Hold* find_hold(Reg& g, const void* owner) {
  for (Hold& h : g.holds)
    if (h.owner == owner) return &h;
  return nullptr;
}

// drop `owner`'s hold on entry i; the entry goes when nobody holds it any more.  Returns true when it was erased.
bool drop_hold_locked(Registry& r, size_t i, const void* owner) {
  Reg& g = r.regs[i];
  for (size_t k = 0; k < g.holds.size(); ++k)
    if (g.holds[k].owner == owner) {
      g.holds.erase(g.holds.begin() + k);
      break;
    }
  if (!g.holds.empty()) return false;
  unregister_locked(r, g);
  r.regs.erase(r.regs.begin() + i);
  return true;
}

void evict_lru_locked(Registry& r, const void* owner, int max_holds, uintptr_t keep_lo) {
  for (;;) {
    int held = 0;
    size_t oldest = size_t(-1);
    unsigned long long stamp = ~0ull;
    for (size_t i = 0; i < r.regs.size(); ++i)
      if (const Hold* h = find_hold(r.regs[i], owner)) {
        ++held;
        if (r.regs[i].lo != keep_lo && h->stamp < stamp) { stamp = h->stamp; oldest = i; }
      }
    if (held <= max_holds || oldest == size_t(-1)) return;
    ++r.counters[RPM_PIN_EVICTED];
    drop_hold_locked(r, oldest, owner);
  }
}

}  // namespace

extern "C" {

void* rpm_pin_acquire(const void* owner, const void* ptr, size_t bytes, int max_holds) {
  if (!owner || !ptr || bytes < RPM_PIN_MIN_BYTES) return nullptr;
  const uintptr_t pg = page_size();
  const uintptr_t p = reinterpret_cast<uintptr_t>(ptr);
  const uintptr_t lo = p & ~(pg - 1), hi = (p + bytes + pg - 1) & ~(pg - 1);
  Registry& r = reg();
  std::lock_guard<std::mutex> lock(r.mu);
  const unsigned long long now = ++r.clock;
  // covered by a live registration: share it
  for (Reg& g : r.regs)
    if (g.lo <= lo && hi <= g.hi) {
      if (Hold* h = find_hold(g, owner)) {
        h->stamp = now;
      } else {
        g.holds.push_back(Hold{owner, now});
        ++r.counters[RPM_PIN_SHARED];
        evict_lru_locked(r, owner, max_holds, g.lo);
      }
      for (Reg& q : r.regs)   // (eviction may have moved the entries)
        if (q.lo <= lo && hi <= q.hi) return q.dbase + (p - q.lo);
      return nullptr;
    }
  // partly overlapping registrations: somebody else's pages are never re-registered under them
  uintptr_t ulo = lo, uhi = hi;
  bool overlap = false;
  for (const Reg& g : r.regs)
    if (g.lo < hi && lo < g.hi) {
      overlap = true;
      for (const Hold& h : g.holds)
        if (h.owner != owner) {
          note(r, RPM_PIN_OVERLAP_REFUSED, "registration overlaps pages another engine holds", lo, hi, hipSuccess);
          return nullptr;
        }
      ulo = std::min(ulo, g.lo);
      uhi = std::max(uhi, g.hi);
    }
  if (overlap) {
    // only this owner's: arrays that share an edge page, or an array the caller re-allocated.  Replace them by their union;
    // if the union is not registrable (part of an old array is gone) by the new array's pages alone.
    for (size_t i = 0; i < r.regs.size();)
      if (r.regs[i].lo < hi && lo < r.regs[i].hi) {
        if (!drop_hold_locked(r, i, owner)) ++i;
      } else {
        ++i;
      }
    ++r.counters[RPM_PIN_MERGED];
  }
  char* dbase = nullptr;
  uintptr_t rlo = ulo, rhi = uhi;
  if (!register_locked(r, rlo, rhi, &dbase)) {
    if (!overlap || (ulo == lo && uhi == hi)) return nullptr;
    rlo = lo;
    rhi = hi;
    if (!register_locked(r, rlo, rhi, &dbase)) return nullptr;
  }
  Reg g;
  g.lo = rlo;
  g.hi = rhi;
  g.dbase = dbase;
  g.holds.push_back(Hold{owner, now});
  r.regs.push_back(g);
  evict_lru_locked(r, owner, max_holds, rlo);
  return dbase + (p - rlo);
}

void rpm_pin_release_owner(const void* owner) {
  if (!owner) return;
  Registry& r = reg();
  std::lock_guard<std::mutex> lock(r.mu);
  for (size_t i = 0; i < r.regs.size();) {
    if (find_hold(r.regs[i], owner)) {
      if (drop_hold_locked(r, i, owner)) continue;
    }
    ++i;
  }
}

int rpm_pin_release_range(const void* owner, const void* ptr, size_t bytes) {
  if (!owner || !ptr) return 0;
  const uintptr_t p = reinterpret_cast<uintptr_t>(ptr);
  Registry& r = reg();
  std::lock_guard<std::mutex> lock(r.mu);
  int n = 0;
  for (size_t i = 0; i < r.regs.size();) {
    if (r.regs[i].lo < p + bytes && p < r.regs[i].hi && find_hold(r.regs[i], owner)) {
      ++n;
      if (drop_hold_locked(r, i, owner)) continue;
    }
    ++i;
  }
  return n;
}

long rpm_pin_counter(int which) {
  Registry& r = reg();
  std::lock_guard<std::mutex> lock(r.mu);
  if (which == RPM_PIN_LIVE) return long(r.regs.size());
  if (which < 0 || which >= RPM_PIN_N_COUNTERS) return -1;
  return r.counters[which];
}

int rpm_pin_held(const void* owner) {
  Registry& r = reg();
  std::lock_guard<std::mutex> lock(r.mu);
  int n = 0;
  for (Reg& g : r.regs)
    if (find_hold(g, owner)) ++n;
  return n;
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

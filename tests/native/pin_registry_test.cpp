// CPU test of the page-lock registry's bookkeeping (lpopc_amd/csrc/rpm_pin.cpp) against a mock of the three runtime calls
// it makes: the mock fails the test on anything that would confuse the real runtime's table or leave memory pinned behind
// a caller's back (profiles/r03_host_register_probe.log: the runtime itself accepts all of it) — registering a byte that is
// registered already, unregistering an unknown base.  Random sequences of
// requests from several owners, overlapping arrays, evictions and releases; after every step the registered
// regions must be exactly the pages somebody still addresses (regions several owners share may be larger: frozen).
// Compiled and run by tests/test_pin_registry_cpu.py.
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <vector>

#define hipHostRegister mock_hipHostRegister
#define hipHostUnregister mock_hipHostUnregister
#define hipHostGetDevicePointer mock_hipHostGetDevicePointer
#include "../../lpopc_amd/csrc/rpm_pin.cpp"

static std::map<uintptr_t, uintptr_t> g_live;   // lo -> hi
static long g_fail_at = -1, g_calls = 0;

#define CHECK(c)                                                         \
  do {                                                                   \
    if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); std::exit(1); } \
  } while (0)

hipError_t mock_hipHostRegister(void* p, size_t bytes, unsigned flags) {
  const uintptr_t lo = reinterpret_cast<uintptr_t>(p), hi = lo + bytes;
  CHECK(bytes > 0);
  CHECK(flags == (hipHostRegisterMapped | hipHostRegisterPortable));
  for (auto& kv : g_live) CHECK(!(kv.first < hi && lo < kv.second));   // never pages that are registered already
  if (++g_calls == g_fail_at) return hipErrorOutOfMemory;               // a refusal the registry has to survive
  g_live[lo] = hi;
  return hipSuccess;
}
hipError_t mock_hipHostUnregister(void* p) {
  auto it = g_live.find(reinterpret_cast<uintptr_t>(p));
  CHECK(it != g_live.end());                                            // only bases that were registered
  g_live.erase(it);
  return hipSuccess;
}
hipError_t mock_hipHostGetDevicePointer(void** d, void* p, unsigned) {
  *d = reinterpret_cast<char*>(p) + (1ull << 44);                       // a device alias that is not the host address
  return hipSuccess;
}
// the two runtime functions rpm_pin.cpp calls besides (error text, sticky-error reset)
extern "C" const char* hipGetErrorString(hipError_t) { return "mock refusal"; }
extern "C" hipError_t hipGetLastError(void) { return hipSuccess; }

struct Held { uintptr_t lo, hi; };

int main() {
  std::mt19937_64 rng(12345);
  const uintptr_t base = 0x7f0000000000ull;
  const int OWNERS = 4, CAP = 8;
  int owners[OWNERS];
  long served = 0, refused = 0;
  for (int round = 0; round < 20000; ++round) {
    const int o = int(rng() % OWNERS);
    const int op = int(rng() % 100);
    if (op < 90) {
      // arrays from a small arena so that they collide: sizes 64 KB .. 1 MB, byte-granular starts
      const uintptr_t start = base + (rng() % (24u << 20));
      const size_t bytes = 65536 + rng() % (1u << 20);
      if (round % 997 == 0) g_fail_at = g_calls + 1;
      void* alias = rpm_pin_acquire(&owners[o], reinterpret_cast<void*>(start), bytes, CAP, 0);
      g_fail_at = -1;
      if (alias) {
        ++served;
        CHECK(reinterpret_cast<uintptr_t>(alias) == start + (1ull << 44));
        bool covered = false;                                          // the whole array lies in ONE registered region
        for (auto& kv : g_live) covered |= (kv.first <= start && start + bytes <= kv.second);
        CHECK(covered);
        // asking again is a pure lookup: same alias, no runtime call
        const long calls = g_calls;
        CHECK(rpm_pin_acquire(&owners[o], reinterpret_cast<void*>(start), bytes, CAP, 0) == alias && g_calls == calls);
      } else {
        ++refused;
      }
    } else if (op < 97) {
      rpm_pin_release_owner(&owners[o]);
      CHECK(rpm_pin_held(&owners[o]) == 0);
    } else {
      CHECK(rpm_pin_acquire(&owners[o], reinterpret_cast<void*>(base), 65535, CAP, 0) == nullptr);   // below the threshold
      void* small = rpm_pin_acquire(&owners[o], reinterpret_cast<void*>(base + (40u << 20)), 4096, CAP, 1);   // ... unless asked for
      CHECK(small == nullptr || reinterpret_cast<uintptr_t>(small) == base + (40u << 20) + (1ull << 44));
    }
    long bytes_live = 0;
    for (auto& kv : g_live) bytes_live += long(kv.second - kv.first);
    CHECK(rpm_pin_counter(RPM_PIN_LIVE) == long(g_live.size()) && rpm_pin_counter(RPM_PIN_LIVE_BYTES) == bytes_live);
    for (int k = 0; k < OWNERS; ++k) CHECK(rpm_pin_held(&owners[k]) <= CAP);
    // every registered region holds at least one array, and no owner's arrays are left without a region
    Registry& r = reg();
    for (const Reg& g : r.regs) {
      bool any = false;
      for (const Arr& a : r.arrs) any |= (g.lo <= a.lo && a.hi <= g.hi);
      CHECK(any);
      int owners_in = 0;
      const void* first = nullptr;
      uintptr_t ulo = ~uintptr_t(0), uhi = 0;
      for (const Arr& a : r.arrs)
        if (g.lo <= a.lo && a.hi <= g.hi) {
          if (!first) { first = a.owner; owners_in = 1; }
          else if (a.owner != first) owners_in = 2;
          ulo = std::min(ulo, a.lo);
          uhi = std::max(uhi, a.hi);
        }
      if (owners_in == 1 && !g.was_shared) CHECK(ulo == g.lo && uhi == g.hi);   // one owner from the start: exactly its arrays' pages
    }
    for (const Arr& a : r.arrs) CHECK(region_of(r, a.lo, a.hi) != nullptr);
    CHECK(rpm_pin_counter(RPM_PIN_UNREGISTER_FAILURES) == 0);
  }
  for (int k = 0; k < OWNERS; ++k) rpm_pin_release_owner(&owners[k]);
  CHECK(g_live.empty() && rpm_pin_counter(RPM_PIN_LIVE) == 0);
  CHECK(rpm_pin_counter(RPM_PIN_REGISTERED) == rpm_pin_counter(RPM_PIN_UNREGISTERED));
  char msg[256];
  rpm_pin_last_error(msg, sizeof msg);
  CHECK(rpm_pin_counter(RPM_PIN_REGISTER_FAILURES) > 0 && rpm_pin_counter(RPM_PIN_OVERLAP_REFUSED) > 0 && msg[0] != 0);
  std::printf("ok: %ld served, %ld refused, %ld registered, %ld shared, %ld merged, %ld evicted, %ld overlap-refused, %ld runtime refusals\n",
              served, refused, rpm_pin_counter(RPM_PIN_REGISTERED), rpm_pin_counter(RPM_PIN_SHARED), rpm_pin_counter(RPM_PIN_MERGED),
              rpm_pin_counter(RPM_PIN_EVICTED), rpm_pin_counter(RPM_PIN_OVERLAP_REFUSED), rpm_pin_counter(RPM_PIN_REGISTER_FAILURES));
  return 0;
}

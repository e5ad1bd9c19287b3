// host_register_probe.hip — what the HIP runtime does with host pointers near a hipHostRegister'ed range (the facts the
// page-lock registry, lpopc_amd/csrc/rpm_pin.cpp, is built on).  Build: hipcc --offload-arch=gfx950 -o probe host_register_probe.hip
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>

#define SHOW(label, call) do { std::printf("%-86s -> ", label); std::fflush(stdout); hipError_t s_ = (call); std::printf("%s\n", s_ == hipSuccess ? "ok" : hipGetErrorString(s_)); std::fflush(stdout); (void)hipGetLastError(); } while (0)

int main(int argc, char** argv) {
  const bool stale_case = argc > 1 && !std::strcmp(argv[1], "--stale");   // case 4 ends the process with a GPU memory fault: off by default
  const size_t PG = 4096, N = 64 * PG;
  char* base = static_cast<char*>(mmap(nullptr, 4 * N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
  std::memset(base, 1, 4 * N);
  void* dev = nullptr;
  SHOW("hipMalloc", hipMalloc(&dev, 4 * N));
  // 1. two byte ranges that share a page
  char* a = base + 100;             // [100, 100 + N)            ends inside page 64
  char* b = a + N + 8;              // starts 8 bytes later, same page
  SHOW("1a register A = [100, 100+N)", hipHostRegister(a, N, hipHostRegisterMapped));
  SHOW("1b register B = 8 bytes after A's end (shares a page with A)", hipHostRegister(b, N, hipHostRegisterMapped));
  SHOW("1c register A again (same pointer, same size)", hipHostRegister(a, N, hipHostRegisterMapped));
  SHOW("1d register a range inside A", hipHostRegister(a + PG, PG, hipHostRegisterMapped));
  SHOW("1e register a range that starts inside A and ends beyond", hipHostRegister(a + N - PG, 2 * PG, hipHostRegisterMapped));
  // 2. pageable-looking copies near the registered range
  SHOW("2a copy H2D from inside A, fully covered", hipMemcpy(dev, a + 256, PG, hipMemcpyHostToDevice));
  SHOW("2b copy H2D that starts inside A and ends beyond its last byte (into B's start)", hipMemcpy(dev, a + N - 64, 256, hipMemcpyHostToDevice));
  SHOW("2c copy H2D that starts before A (first 100 bytes of its first page) and runs into A", hipMemcpy(dev, base, 4096, hipMemcpyHostToDevice));
  SHOW("2d copy H2D from the unregistered first 64 bytes of A's first page", hipMemcpy(dev, base, 64, hipMemcpyHostToDevice));
  SHOW("2e copy D2H to a range that starts inside A and ends beyond", hipMemcpy(a + N - 64, dev, 256, hipMemcpyDeviceToHost));
  SHOW("unregister B", hipHostUnregister(b));
  SHOW("unregister A", hipHostUnregister(a));
  SHOW("unregister A a second time", hipHostUnregister(a));
  // 3. page-aligned superset: neighbours on the edge pages
  char* c = base + 2 * N + 100;
  char* clo = base + 2 * N;         // page-aligned superset [2N, 2N + N + PG)
  SHOW("3a register the page-aligned superset of C", hipHostRegister(clo, N + PG, hipHostRegisterMapped));
  SHOW("3b copy H2D from a neighbour object that starts on C's last page and ends beyond the superset", hipMemcpy(dev, clo + N + PG - 64, 256, hipMemcpyHostToDevice));
  SHOW("3c copy H2D of C itself", hipMemcpy(dev, c, N, hipMemcpyHostToDevice));
  if (!stale_case) { SHOW("unregister superset", hipHostUnregister(clo)); std::printf("probe finished (case 4 not run)\n"); return 0; }
  // 4. unmap a registered range, map something else there, copy from it (what a freed-and-reallocated numpy array is)
  char* d = static_cast<char*>(mmap(nullptr, N, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
  std::memset(d, 2, N);
  SHOW("4a register D (own mapping)", hipHostRegister(d, N, hipHostRegisterMapped));
  munmap(d, N);
  char* d2 = static_cast<char*>(mmap(d, N / 2, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_FIXED, -1, 0));
  std::memset(d2, 3, N / 2);
  std::printf("   D unmapped while registered; a new, smaller mapping placed at the same address (%s)\n", d2 == d ? "same address" : "moved");
  SHOW("4b copy H2D from the new mapping (inside the stale registration)", hipMemcpy(dev, d2, N / 2, hipMemcpyHostToDevice));
  char chk[16];
  SHOW("4c read back", hipMemcpy(chk, dev, 16, hipMemcpyDeviceToHost));
  std::printf("   device received bytes of value %d (3 = the new mapping's contents, 2 = the old pages)\n", int(chk[0]));
  SHOW("4d unregister the stale D", hipHostUnregister(d));
  SHOW("unregister superset", hipHostUnregister(clo));
  std::printf("probe finished\n");
  return 0;
}

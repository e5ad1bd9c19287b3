// pcie_paths.hip — what each piece of the host-pointer (Ipopt-facing) path costs on this box, measured with the
// host's clock around "queue + complete" (what rpm_eval_g / rpm_eval_jac_g see), median of many repetitions.
// Sizes are the metric problem's: x 40 996, g 32 801, NL Jacobian prefix 393 604, all values 852 356 doubles.
//   build: hipcc --offload-arch=gfx950 -O3 -o pcie_paths.bin pcie_paths.hip        run: ./pcie_paths.bin > out.jsonl
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t s_ = (x); if (s_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(s_)); exit(1); } } while (0)

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

__global__ void k_empty() {}
// linear copy, 16 B per lane
__global__ __launch_bounds__(256) void k_copy16(const double2* __restrict__ src, double2* __restrict__ dst, size_t n2) {
  for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n2; i += size_t(gridDim.x) * 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void k_copy8(const double* __restrict__ src, double* __restrict__ dst, size_t n) {
  for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += size_t(gridDim.x) * 256) dst[i] = src[i];
}
// the one-role tile kernel's store pattern: a workgroup owns 16 consecutive nodes, thread = (block b, node), each of the
// nblk blocks is N doubles long: 128-byte runs at a stride of 8 N bytes
__global__ __launch_bounds__(192) void k_tile_pattern(const double* __restrict__ src, double* __restrict__ dst, int N, int nblk) {
  const int node = blockIdx.x * 16 + (threadIdx.x & 15);
  for (int b = threadIdx.x >> 4; b < nblk; b += 12) dst[size_t(b) * N + node] = src[size_t(b) * N + node];
}
// the same with 64 nodes per workgroup (512-byte runs)
__global__ __launch_bounds__(256) void k_tile_pattern64(const double* __restrict__ src, double* __restrict__ dst, int N, int nblk) {
  const int node = blockIdx.x * 64 + (threadIdx.x & 63);
  for (int b = threadIdx.x >> 6; b < nblk; b += 4) dst[size_t(b) * N + node] = src[size_t(b) * N + node];
}
// completion flag in host memory instead of hipStreamSynchronize: every workgroup drains its stores, the last one to
// arrive (device counter) publishes `epoch` to the host word with a system-scope release
__global__ __launch_bounds__(256) void k_copy16_flag(const double2* __restrict__ src, double2* __restrict__ dst, size_t n2,
                                                     unsigned* counter, volatile unsigned* host_flag, unsigned epoch) {
  for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n2; i += size_t(gridDim.x) * 256) dst[i] = src[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned prev = atomicAdd(counter, 1u);
    if (prev == gridDim.x - 1) {
      *counter = 0;
      __threadfence_system();
      __hip_atomic_store(const_cast<unsigned*>(host_flag), epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

template <class F>
static void measure(const char* name, size_t bytes, int reps, F&& body, const char* note = "") {
  std::vector<double> t;
  for (int r = 0; r < reps + 5; ++r) {
    const double a = now_us();
    body();
    const double b = now_us();
    if (r >= 5) t.push_back(b - a);
  }
  std::sort(t.begin(), t.end());
  const double med = t[t.size() / 2], p10 = t[t.size() / 10], p90 = t[t.size() * 9 / 10];
  printf("{\"what\": \"%s\", \"bytes\": %zu, \"median_us\": %.2f, \"p10_us\": %.2f, \"p90_us\": %.2f, \"GBps_at_median\": %.2f, \"note\": \"%s\"}\n",
         name, bytes, med, p10, p90, bytes ? bytes / med * 1e-3 : 0.0, note);
  fflush(stdout);
}

int main() {
  const size_t N_X = 40996, N_G = 32801, N_NL = 393604, N_ALL = 852356;
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  double *d_src, *d_dst;
  CK(hipMalloc(&d_src, N_ALL * 8 + 4096));
  CK(hipMalloc(&d_dst, N_ALL * 8 + 4096));
  CK(hipMemset(d_src, 1, N_ALL * 8));
  double* h_pin;   // hipHostMalloc: page-locked, mapped, coherent
  CK(hipHostMalloc(&h_pin, N_ALL * 8 + 4096, hipHostMallocMapped));
  memset(h_pin, 0, N_ALL * 8);
  double* h_reg = static_cast<double*>(aligned_alloc(4096, N_ALL * 8 + 4096));   // the caller's malloc'ed array, registered
  memset(h_reg, 0, N_ALL * 8);
  CK(hipHostRegister(h_reg, N_ALL * 8 + 4096, hipHostRegisterMapped));
  double *hd_pin, *hd_reg;
  CK(hipHostGetDevicePointer(reinterpret_cast<void**>(&hd_pin), h_pin, 0));
  CK(hipHostGetDevicePointer(reinterpret_cast<void**>(&hd_reg), h_reg, 0));
  double* h_page = static_cast<double*>(malloc(N_ALL * 8));
  memset(h_page, 0, N_ALL * 8);
  unsigned* d_counter;
  CK(hipMalloc(&d_counter, 64));
  CK(hipMemset(d_counter, 0, 64));
  unsigned* h_flag;
  CK(hipHostMalloc(&h_flag, 64, hipHostMallocMapped));
  *h_flag = 0;
  unsigned* hd_flag;
  CK(hipHostGetDevicePointer(reinterpret_cast<void**>(&hd_flag), h_flag, 0));
  CK(hipDeviceSynchronize());
  const int R = 300;

  measure("empty kernel + hipStreamSynchronize", 0, R, [&] {
    hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st);
    CK(hipStreamSynchronize(st));
  });
  measure("H2D x, hipMemcpyAsync from registered + sync", N_X * 8, R, [&] {
    CK(hipMemcpyAsync(d_dst, h_reg, N_X * 8, hipMemcpyHostToDevice, st));
    CK(hipStreamSynchronize(st));
  });
  measure("H2D x, hipMemcpyAsync from pageable + sync", N_X * 8, R, [&] {
    CK(hipMemcpyAsync(d_dst, h_page, N_X * 8, hipMemcpyHostToDevice, st));
    CK(hipStreamSynchronize(st));
  });
  measure("x read by a kernel from mapped host memory (copy to HBM) + sync", N_X * 8, R, [&] {
    hipLaunchKernelGGL(k_copy8, dim3(160), dim3(256), 0, st, hd_reg, d_dst, N_X);
    CK(hipStreamSynchronize(st));
  });
  for (size_t n : {N_G, N_NL, N_ALL}) {
    measure("D2H hipMemcpyAsync to registered + sync", n * 8, R, [&] {
      CK(hipMemcpyAsync(h_reg, d_src, n * 8, hipMemcpyDeviceToHost, st));
      CK(hipStreamSynchronize(st));
    });
    measure("D2H hipMemcpyAsync to hipHostMalloc + sync", n * 8, R, [&] {
      CK(hipMemcpyAsync(h_pin, d_src, n * 8, hipMemcpyDeviceToHost, st));
      CK(hipStreamSynchronize(st));
    });
  }
  measure("D2H hipMemcpyAsync to pageable + sync", N_NL * 8, 100, [&] {
    CK(hipMemcpyAsync(h_page, d_src, N_NL * 8, hipMemcpyDeviceToHost, st));
    CK(hipStreamSynchronize(st));
  });
  for (size_t n : {N_G, N_NL, N_ALL}) {
    for (int wgs : {16, 32, 64, 128, 256, 512, 1024}) {
      char note[64];
      snprintf(note, sizeof note, "%d workgroups", wgs);
      measure("kernel copy HBM -> registered host, 16 B/lane + sync", n * 8, R, [&] {
        hipLaunchKernelGGL(k_copy16, dim3(wgs), dim3(256), 0, st, reinterpret_cast<const double2*>(d_src),
                           reinterpret_cast<double2*>(hd_reg), n / 2);
        CK(hipStreamSynchronize(st));
      }, note);
    }
  }
  measure("kernel copy HBM -> registered host, 8 B/lane, 256 wgs + sync", N_NL * 8, R, [&] {
    hipLaunchKernelGGL(k_copy8, dim3(256), dim3(256), 0, st, d_src, hd_reg, N_NL);
    CK(hipStreamSynchronize(st));
  });
  {
    // NL prefix as 96 blocks of N = 4100 (rounded: 256 tiles of 16 nodes = 4096 nodes, 96 blocks -> 393 216 doubles)
    const int N = 4096, nblk = 96;
    measure("tile store pattern (128-B runs, stride 32 KB) HBM -> registered host + sync", size_t(N) * nblk * 8, R, [&] {
      hipLaunchKernelGGL(k_tile_pattern, dim3(N / 16), dim3(192), 0, st, d_src, hd_reg, N, nblk);
      CK(hipStreamSynchronize(st));
    });
    measure("tile store pattern (512-B runs, stride 32 KB) HBM -> registered host + sync", size_t(N) * nblk * 8, R, [&] {
      hipLaunchKernelGGL(k_tile_pattern64, dim3(N / 64), dim3(256), 0, st, d_src, hd_reg, N, nblk);
      CK(hipStreamSynchronize(st));
    });
    measure("tile store pattern (128-B runs) HBM -> HBM + sync (for reference)", size_t(N) * nblk * 8, R, [&] {
      hipLaunchKernelGGL(k_tile_pattern, dim3(N / 16), dim3(192), 0, st, d_src, d_dst, N, nblk);
      CK(hipStreamSynchronize(st));
    });
  }
  // completion through a host flag the kernel writes, polled by the CPU, instead of hipStreamSynchronize
  unsigned epoch = 0;
  for (size_t n : {size_t(0), N_G, N_NL}) {
    measure("kernel copy -> registered host, 128 wgs, completion by host-polled flag", n * 8, R, [&] {
      ++epoch;
      hipLaunchKernelGGL(k_copy16_flag, dim3(128), dim3(256), 0, st, reinterpret_cast<const double2*>(d_src),
                         reinterpret_cast<double2*>(hd_reg), n / 2, d_counter, hd_flag, epoch);
      while (__atomic_load_n(h_flag, __ATOMIC_ACQUIRE) != epoch) {}
    });
  }
  // SDMA copy of one half beside a kernel copy of the other half (two streams)
  hipStream_t st2;
  CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
  for (int pct : {25, 50, 75}) {
    char note[64];
    snprintf(note, sizeof note, "%d %% by the copy engine", pct);
    const size_t a = (N_NL * pct / 100) & ~size_t(1), b = N_NL - a;
    measure("D2H split: hipMemcpyAsync part + kernel-copy part, two streams", N_NL * 8, R, [&] {
      CK(hipMemcpyAsync(h_reg, d_src, a * 8, hipMemcpyDeviceToHost, st));
      hipLaunchKernelGGL(k_copy16, dim3(128), dim3(256), 0, st2, reinterpret_cast<const double2*>(d_src + a),
                         reinterpret_cast<double2*>(hd_reg + a), b / 2);
      CK(hipStreamSynchronize(st));
      CK(hipStreamSynchronize(st2));
    }, note);
  }
  // the whole pair as ONE zero-copy kernel: x read from host, g + NL prefix stored to host, one completion
  measure("pair emulation: kernel reads x from host, stores g + NL to host, flag completion", (N_X + N_G + N_NL) * 8, R, [&] {
    ++epoch;
    hipLaunchKernelGGL(k_copy8, dim3(160), dim3(256), 0, st, hd_pin, d_dst, N_X);
    hipLaunchKernelGGL(k_copy16_flag, dim3(128), dim3(256), 0, st, reinterpret_cast<const double2*>(d_src),
                       reinterpret_cast<double2*>(hd_reg), (N_G + N_NL) / 2, d_counter, hd_flag, epoch);
    while (__atomic_load_n(h_flag, __ATOMIC_ACQUIRE) != epoch) {}
  });
  measure("pair emulation, hipStreamSynchronize completion", (N_X + N_G + N_NL) * 8, R, [&] {
    hipLaunchKernelGGL(k_copy8, dim3(160), dim3(256), 0, st, hd_pin, d_dst, N_X);
    hipLaunchKernelGGL(k_copy16, dim3(128), dim3(256), 0, st, reinterpret_cast<const double2*>(d_src),
                       reinterpret_cast<double2*>(hd_reg), (N_G + N_NL) / 2);
    CK(hipStreamSynchronize(st));
  });
  // host memcpy rate (what an engine-owned staging buffer would add)
  measure("host memcpy pinned staging -> caller array (1 thread)", N_NL * 8, 100, [&] { memcpy(h_page, h_pin, N_NL * 8); });
  // correctness of the flag protocol: the data the flag announces is complete
  {
    CK(hipMemset(d_src, 0, N_NL * 8));
    int bad = 0;
    for (int it = 1; it <= 50; ++it) {
      std::vector<double> pat(N_NL, double(it));
      CK(hipMemcpy(d_src, pat.data(), N_NL * 8, hipMemcpyHostToDevice));
      ++epoch;
      hipLaunchKernelGGL(k_copy16_flag, dim3(128), dim3(256), 0, st, reinterpret_cast<const double2*>(d_src),
                         reinterpret_cast<double2*>(hd_reg), N_NL / 2, d_counter, hd_flag, epoch);
      while (__atomic_load_n(h_flag, __ATOMIC_ACQUIRE) != epoch) {}
      for (size_t i = 0; i < N_NL; ++i) bad += h_reg[i] != double(it);
      CK(hipStreamSynchronize(st));
    }
    printf("{\"what\": \"flag protocol check: stale doubles seen after the flag over 50 transfers\", \"bad\": %d}\n", bad);
  }
  return 0;
}

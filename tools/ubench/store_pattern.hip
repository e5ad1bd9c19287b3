// Store-pattern microbenchmark (perf exploration): how fast can 256-thread workgroups write the Jacobian's layout?
// mode 0: linear, 8 B per lane (512 B per wave instruction)
// mode 1: the tile kernel's pattern: workgroup = 64 nodes, wave w writes runs (blk = w, w+4, ...) of 512 B at stride N*8
// mode 2: linear, 16 B per lane
// mode 3: like 1 but 128 nodes per workgroup and 16 B per lane (1 KB runs)
#include <hip/hip_runtime.h>
extern "C" {
__global__ __launch_bounds__(256) void k_linear8(double* out, size_t per_wg) {
  double* p = out + size_t(blockIdx.x) * per_wg;
  for (size_t q = threadIdx.x; q < per_wg; q += 256) p[q] = 1.0 + q;
}
__global__ __launch_bounds__(256) void k_linear16(double2* out, size_t per_wg2) {
  double2* p = out + size_t(blockIdx.x) * per_wg2;
  for (size_t q = threadIdx.x; q < per_wg2; q += 256) p[q] = make_double2(1.0 + q, 2.0);
}
// nblk blocks of N doubles per instance; tiles of 64 nodes; grid.x = tiles per instance * instances
__global__ __launch_bounds__(256) void k_pattern8(double* out, int N, int nblk, size_t inst_stride) {
  const int tiles = N / 64;
  const int inst = blockIdx.x / tiles, tile = blockIdx.x % tiles;
  double* p = out + size_t(inst) * inst_stride + tile * 64 + (threadIdx.x & 63);
  for (int b = threadIdx.x >> 6; b < nblk; b += 4) p[size_t(b) * N] = 1.0 + b;
}
__global__ __launch_bounds__(256) void k_pattern8_nt(double* out, int N, int nblk, size_t inst_stride) {
  const int tiles = N / 64;
  const int inst = blockIdx.x / tiles, tile = blockIdx.x % tiles;
  double* p = out + size_t(inst) * inst_stride + tile * 64 + (threadIdx.x & 63);
  for (int b = threadIdx.x >> 6; b < nblk; b += 4) __builtin_nontemporal_store(1.0 + b, p + size_t(b) * N);
}
__global__ __launch_bounds__(256) void k_linear16_nt(double2* out, size_t per_wg2) {
  double* p = reinterpret_cast<double*>(out + size_t(blockIdx.x) * per_wg2);
  for (size_t q = threadIdx.x; q < per_wg2; q += 256) {
    __builtin_nontemporal_store(1.0 + q, p + 2 * q);
    __builtin_nontemporal_store(2.0, p + 2 * q + 1);
  }
}
__global__ __launch_bounds__(256) void k_pattern16(double2* out, int N, int nblk, size_t inst_stride) {
  const int tiles = N / 128;
  const int inst = blockIdx.x / tiles, tile = blockIdx.x % tiles;
  double2* p = out + (size_t(inst) * inst_stride + tile * 128) / 2 + (threadIdx.x & 63);
  for (int b = threadIdx.x >> 6; b < nblk; b += 4) p[size_t(b) * N / 2] = make_double2(1.0 + b, 2.0);
}
// returns average microseconds per launch over `reps` launches on `stream`
float run(int mode, int lds_bytes, void** outs, int nbuf, int N, int nblk, int instances, size_t inst_stride, int reps, void* stream) {
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (lds_bytes > 64 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_linear8), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_pattern8), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_linear16), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_pattern16), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const size_t total = size_t(instances) * nblk * N;
  for (int r = 0; r < reps + 3; ++r) {
    void* out = outs[r % nbuf];
    if (r == 3) (void)hipEventRecord(e0, st);
    if (mode == 0) hipLaunchKernelGGL(k_linear8, dim3(instances * (N / 64)), dim3(256), lds_bytes, st, static_cast<double*>(out), size_t(64) * nblk);
    if (mode == 2) hipLaunchKernelGGL(k_linear16, dim3(instances * (N / 64)), dim3(256), lds_bytes, st, static_cast<double2*>(out), size_t(32) * nblk);
    if (mode == 1) hipLaunchKernelGGL(k_pattern8, dim3(instances * (N / 64)), dim3(256), lds_bytes, st, static_cast<double*>(out), N, nblk, inst_stride);
    if (mode == 4) hipLaunchKernelGGL(k_pattern8_nt, dim3(instances * (N / 64)), dim3(256), lds_bytes, st, static_cast<double*>(out), N, nblk, inst_stride);
    if (mode == 5) hipLaunchKernelGGL(k_linear16_nt, dim3(instances * (N / 64)), dim3(256), lds_bytes, st, static_cast<double2*>(out), size_t(32) * nblk);
    if (mode == 3) hipLaunchKernelGGL(k_pattern16, dim3(instances * (N / 128)), dim3(256), lds_bytes, st, static_cast<double2*>(out), N, nblk, inst_stride);
  }
  (void)hipEventRecord(e1, st);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)total;
  return ms * 1e3f / reps;
}
}

// Issue rate / completion time of global_load_lds_dwordx4 (perf exploration): one wave (or several) per workgroup copies
// NCH chunks of 1 KB from an L2-resident buffer into LDS; stamps: start, all issued, all landed (s_waitcnt vmcnt(0)).
#include <hip/hip_runtime.h>
#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))
template <int MODE>
__device__ void body(const double* src, unsigned long long* stamps, int nch, int waves) {
  extern __shared__ double sm[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const double* run = src + size_t(blockIdx.x % 64) * 4096;
  __syncthreads();
  unsigned long long t0 = wall_clock64();
  if (wv < waves) {
    if (MODE == 0) {   // direct-to-LDS, new M0 per chunk
      for (int ch = wv; ch < nch; ch += waves)
        __builtin_amdgcn_global_load_lds(GPTR(run + ch * 128 + 2 * lane), LPTR(sm + ch * 128), 16, 0, 0);
    } else if (MODE == 1) {   // direct-to-LDS, dword variant (256 B per instruction), same bytes
      for (int ch = wv; ch < nch * 4; ch += waves)
        __builtin_amdgcn_global_load_lds(GPTR(reinterpret_cast<const int*>(run) + ch * 64 + lane), LPTR(reinterpret_cast<int*>(sm) + ch * 64), 4, 0, 0);
    } else {   // ordinary loads into registers, then LDS writes
      typedef double d2 __attribute__((ext_vector_type(2), aligned(8)));
      d2 r[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) { const int ch = wv + q * waves; r[q] = ch < nch ? *reinterpret_cast<const d2*>(run + ch * 128 + 2 * lane) : d2{0, 0}; }
#pragma unroll
      for (int q = 0; q < 8; ++q) { const int ch = wv + q * waves; if (ch < nch) *reinterpret_cast<d2*>(sm + ch * 128 + 2 * lane) = r[q]; }
    }
  }
  unsigned long long t1 = wall_clock64();
  __builtin_amdgcn_s_waitcnt(0);
  unsigned long long t2 = wall_clock64();
  __syncthreads();
  if (threadIdx.x == 0) { stamps[blockIdx.x * 4] = t0; stamps[blockIdx.x * 4 + 1] = t1; stamps[blockIdx.x * 4 + 2] = t2; stamps[blockIdx.x * 4 + 3] = (unsigned long long)sm[5]; }
}
extern "C" {
__global__ void k0(const double* s, unsigned long long* st, int nch, int waves) { body<0>(s, st, nch, waves); }
__global__ void k1(const double* s, unsigned long long* st, int nch, int waves) { body<1>(s, st, nch, waves); }
__global__ void k2(const double* s, unsigned long long* st, int nch, int waves) { body<2>(s, st, nch, waves); }
void run(int mode, const double* src, unsigned long long* stamps, int blocks, int nch, int waves) {
  for (int r = 0; r < 3; ++r) {
    if (mode == 0) hipLaunchKernelGGL(k0, dim3(blocks), dim3(256), 64 * 1024, nullptr, src, stamps, nch, waves);
    if (mode == 1) hipLaunchKernelGGL(k1, dim3(blocks), dim3(256), 64 * 1024, nullptr, src, stamps, nch, waves);
    if (mode == 2) hipLaunchKernelGGL(k2, dim3(blocks), dim3(256), 64 * 1024, nullptr, src, stamps, nch, waves);
  }
  (void)hipDeviceSynchronize();
}
}

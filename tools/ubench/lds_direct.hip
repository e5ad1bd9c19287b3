// Semantics check of the gfx950 direct-to-LDS loads (global_load_lds_dwordx4 / dword), perf exploration.
#include <hip/hip_runtime.h>
#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))
extern "C" {
// copies src[off .. off+len) into LDS at sm[8 .. 8+len) with one wave, 16 B per lane, tail by dwords; then out = sm
__global__ void k_copy(const double* src, int off, int len, double* out, int nout) {
  extern __shared__ double sm[];
  const int lane = threadIdx.x & 63;
  for (int q = threadIdx.x; q < nout; q += blockDim.x) sm[q] = -1.0;
  __syncthreads();
  if (threadIdx.x < 64) {
    const double* run = src + off;
    const int pairs = len / 2;
    for (int ch = 0; ch * 64 < pairs; ++ch) {
      if (ch * 64 + lane < pairs)
        __builtin_amdgcn_global_load_lds(GPTR(run + ch * 128 + 2 * lane), LPTR(sm + 9 + ch * 128), 16, 0, 0);
    }
    if ((len & 1) && lane < 2)   // last double as two dwords
      __builtin_amdgcn_global_load_lds(GPTR(reinterpret_cast<const int*>(run + len - 1) + lane), LPTR(sm + 9 + len - 1), 4, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
  }
  __syncthreads();
  for (int q = threadIdx.x; q < nout; q += blockDim.x) out[q] = sm[q];
}
}

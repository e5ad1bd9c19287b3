// How much store bandwidth do W waves per CU sustain?  (perf exploration)  Each wave streams its own contiguous region
// with 8 B or 16 B per lane per instruction; grid = waves_per_cu * 256 single-wave workgroups, rotating buffers.
#include <hip/hip_runtime.h>
extern "C" {
__global__ __launch_bounds__(64) void k8(double* out, size_t per_wave) {
  double* p = out + size_t(blockIdx.x) * per_wave;
  for (size_t q = threadIdx.x; q < per_wave; q += 64) p[q] = 1.0 + q;
}
__global__ __launch_bounds__(64) void k16(double2* out, size_t per_wave2) {
  double2* p = out + size_t(blockIdx.x) * per_wave2;
  for (size_t q = threadIdx.x; q < per_wave2; q += 64) p[q] = make_double2(1.0 + q, 2.0);
}
float run(int width, int waves, size_t total_doubles, void** outs, int nbuf, int reps) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const size_t per_wave = total_doubles / waves / 2 * 2;
  for (int r = 0; r < reps + 3; ++r) {
    if (r == 3) (void)hipEventRecord(e0, nullptr);
    if (width == 8) hipLaunchKernelGGL(k8, dim3(waves), dim3(64), 0, nullptr, static_cast<double*>(outs[r % nbuf]), per_wave);
    else hipLaunchKernelGGL(k16, dim3(waves), dim3(64), 0, nullptr, static_cast<double2*>(outs[r % nbuf]), per_wave / 2);
  }
  (void)hipEventRecord(e1, nullptr);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return ms * 1e3f / reps;
}
}

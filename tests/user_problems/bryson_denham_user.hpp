// TEST FIXTURE: the built-in BrysonDenhamProblem functor restated as a user's header (struct rpm::UserProblem), so that a library built
// by lpopc_amd.userproblem.build() can be compared bit for bit with the package's own library (and thereby with the oracle).
#pragma once
#include <hip/hip_runtime.h>
#ifndef RPM_DEV
#define RPM_DEV __device__ __forceinline__
#endif
namespace rpm {
struct UserProblem {
  static constexpr int ID = 100;
  static constexpr int NX = 3, NU = 1, NC = 0, NE_MAX = 5, NLINK_MAX = 0, NCONST = 0;
  static constexpr bool HAS_ANALYTIC = false;
  template <class CP = const double*>
  RPM_DEV static void dae(int, double, const double* x, const double* u, CP, double* f, double*) {
    f[0] = x[1];
    f[1] = u[0];
    f[2] = 0.5 * (u[0] * u[0]);                                        // :121-123
  }
  RPM_DEV static void event(int, double, const double* x0, double, const double* xf, const double*, double* ev) {
    ev[0] = x0[0]; ev[1] = x0[1]; ev[2] = x0[2]; ev[3] = xf[0]; ev[4] = xf[1];  // :139-153
  }
  RPM_DEV static void link(int, int, const double*, const double*, const double*, int, double*) {}
  RPM_DEV static double mayer(int, double, const double*, double, const double* xf, const double*) { return xf[2]; }
  RPM_DEV static double lagrange(int, double, const double*, const double*, const double*) { return 0.0; }
};
}  // namespace rpm

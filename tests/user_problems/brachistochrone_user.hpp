// TEST FIXTURE: the built-in BrachistochroneProblem functor restated as a user's header (struct rpm::UserProblem), so that a library built
// by lpopc_amd.userproblem.build() can be compared bit for bit with the package's own library (and thereby with the oracle).
#pragma once
#include <hip/hip_runtime.h>
#ifndef RPM_DEV
#define RPM_DEV __device__ __forceinline__
#endif
namespace rpm {
struct UserProblem {
  static constexpr int ID = 100;
  static constexpr int NX = 3, NU = 1, NC = 0, NE_MAX = 5, NLINK_MAX = 0, NCONST = 1;
  static constexpr bool HAS_ANALYTIC = true;
  template <class CP = const double*>
  RPM_DEV static void dae(int, double, const double* x, const double* u, CP c, double* f, double*) {
    const double sn = sin(u[0]), cs = cos(u[0]);
    f[0] = x[2] * sn;
    f[1] = x[2] * cs;
    f[2] = c[0] * cs;
  }
  template <class CP = const double*>
  RPM_DEV static void dae_jac_col(int, int v, double, const double* x, const double* u, CP c,
                                  double* df, double*) {
    const double sn = sin(u[0]), cs = cos(u[0]);
    df[0] = df[1] = df[2] = 0.0;
    if (v == 2) { df[0] = sn; df[1] = cs; }
    if (v == 3) { df[0] = x[2] * cs; df[1] = -(x[2] * sn); df[2] = -(c[0] * sn); }
  }
  RPM_DEV static void event(int, double, const double* x0, double, const double* xf, const double*, double* ev) {
    ev[0] = x0[0]; ev[1] = x0[1]; ev[2] = x0[2]; ev[3] = xf[0]; ev[4] = xf[1];
  }
  // column v of d event / d [x0.., t0, xf.., tf]
  RPM_DEV static void event_jac_col(int, int v, double, const double*, double, const double*, const double*, double* de) {
    for (int i = 0; i < 5; ++i) de[i] = 0.0;
    if (v < 3) de[v] = 1.0;
    if (v == 4) de[3] = 1.0;
    if (v == 5) de[4] = 1.0;
  }
  RPM_DEV static void link(int, int, const double*, const double*, const double*, int, double*) {}
  RPM_DEV static void link_jac_col(int, int, int, const double*, const double*, const double*, int, double*) {}
  RPM_DEV static double mayer(int, double, const double*, double tf, const double*, const double*) { return tf; }
  RPM_DEV static double mayer_grad_col(int, int v, double, const double*, double, const double*, const double*) {
    return v == 2 * NX + 1 ? 1.0 : 0.0;
  }
  RPM_DEV static double lagrange(int, double, const double*, const double*, const double*) { return 0.0; }
  RPM_DEV static double lagrange_grad_col(int, int, double, const double*, const double*, const double*) { return 0.0; }
};
}  // namespace rpm

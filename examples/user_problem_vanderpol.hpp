// A user's problem for the MI355X engine: the controlled Van der Pol oscillator
//     minimise   integral_0^tf (x1^2 + x2^2 + u^2) dt
//     subject to x1' = (1 - x2^2) x1 - x2 + u,   x2' = x1,   -0.3 <= u <= 1   (bounds are set on the Phase, not here)
// written the way lpopc users write a FunctionWrapper subclass (Lpopc/src/Core/LpFunctionWrapper.h:50-69), but POINTWISE:
// every callback sees one collocation node (dae, lagrange) or the phase's end points (mayer, event), because it runs
// inside the GPU kernels.  `ph` is lpopc's 1-based phase_num_, `c` the problem constants handed to ProblemFunctor
// (here c[0] = weight of u^2).  Build:  lpopc_amd.userproblem.build("examples/user_problem_vanderpol.hpp")
#pragma once
#include <hip/hip_runtime.h>

namespace rpm {

struct UserProblem {
  static constexpr int NX = 2, NU = 1, NC = 0;          // states, controls, path constraints per node
  static constexpr int NE_MAX = 0, NLINK_MAX = 0;       // most events of a phase, most linkage constraints of a pair
  static constexpr int NCONST = 1;                      // problem constants expected in rpm_problem_desc.consts
  static constexpr bool HAS_ANALYTIC = false;           // true: also provide the *_jac_col / *_grad_col callbacks

  // FunctionWrapper::DaeFunction: f = dx/dt, p = path constraints, at ONE node
  template <class CP = const double*>
  __device__ __forceinline__ static void dae(int ph, double t, const double* x, const double* u, CP c, double* f, double* p) {
    (void)ph; (void)t; (void)c; (void)p;
    f[0] = (1.0 - x[1] * x[1]) * x[0] - x[1] + u[0];
    f[1] = x[0];
  }
  // FunctionWrapper::EventFunction / LinkFunction: none in this problem
  __device__ __forceinline__ static void event(int, double, const double*, double, const double*, const double*, double*) {}
  __device__ __forceinline__ static void link(int, int, const double*, const double*, const double*, int, double*) {}
  // FunctionWrapper::MayerCost / LagrangeCost
  __device__ __forceinline__ static double mayer(int, double, const double*, double, const double*, const double*) { return 0.0; }
  __device__ __forceinline__ static double lagrange(int, double, const double* x, const double* u, const double* c) {
    return (x[0] * x[0] + x[1] * x[1]) + c[0] * (u[0] * u[0]);
  }
};

}  // namespace rpm

// rpm_ipm.hpp — row f-2: the caller of the hot path, moved onto the device.  The reference hands the NLP to Ipopt 3.12.3
// (Core/LpNLPSolver.cpp:13-53: tol = "Ipopt-tol", hessian_approximation from the option list); Ipopt is not in the
// reference tree, so what is built here is a restatement of its published algorithm (Waechter & Biegler, Math. Program.
// 106, 2006: primal-dual barrier, fraction-to-the-boundary rule, filter line search, inertia correction) for a batch of
// independent instances of one transcription — the MPC sweep of BASELINE config 5 — with every iterate, multiplier,
// KKT matrix and factor resident in HBM.  Not restated: Ipopt's l1 restoration NLP (a Gauss-Newton feasibility
// restoration built from the same kernels takes its place), second-order corrections, the
// adaptive barrier strategy (monotone Fiacco-McCormick here), scaling.  See DESIGN.md §f-2.
//
// The KKT matrix of a collocation NLP is banded once the unknowns are ordered along time: node k's states, controls,
// slacks and multipliers sit together, a defect row reaches the nodes of its own mesh interval only.  What does not
// fit the band (final states, t0/tf, events, linkages, linear rows) goes into a dense border ("arrow").  IpmPlan holds
// that ordering and, for every Jacobian / Hessian COO entry, its slot in the band + border storage.
#pragma once
#include <string>
#include <vector>

#include "rpm_engine.hpp"

namespace rpm {

struct IpmPlan {
  int n = 0, m = 0, ns = 0, nv = 0;   // variables, rows, slacks (one per inequality row), nv = n + ns
  int Nt = 0, Nb = 0, nb = 0;         // KKT order, banded part, border
  int b = 0, CS = 0;                  // half bandwidth, doubles per stored column (b + 1 + nb, rounded up to even)
  std::vector<int> pos;               // unknown -> position; unknowns: [0,n) x, [n,nv) slacks, [nv,nv+m) multipliers
  std::vector<int> row_slack;         // m: slack index or -1 (equality row)
  std::vector<int> slack_row;         // ns
  std::vector<int> fixed;             // n: 1 where x_l == x_u (kept as identity rows of the KKT system)
  std::vector<int> jac_dst, hes_dst;  // storage offset of every COO entry, -1 = dropped (fixed variable)
  std::vector<int> diag_dst;          // Nt: offset of every unknown's diagonal entry
  std::vector<int> slk_dst;           // ns: offset of the (row, slack) entry (value -1)
  std::vector<int> jt_ptr, jt_ent, jt_row;   // Jacobian by column (deterministic J^T lambda)
  long long storage() const { return (long long)Nt * CS; }
  long long at(int pa, int pb) const {        // pa >= pb
    return (long long)pb * CS + (pa < Nb ? pa - pb : b + 1 + pa - Nb);
  }
};

// rpm_ipm.cpp (host only): ordering, band width and the scatter maps from the engine's layout
int build_ipm_plan(Engine& e, IpmPlan& p, std::string* why);

}  // namespace rpm

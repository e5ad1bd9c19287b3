// rpm_ipm.hpp — row f-2: the caller of the hot path, moved onto the device.  The reference hands the NLP to Ipopt 3.12.3
// (Core/LpNLPSolver.cpp:13-53: tol = "Ipopt-tol", hessian_approximation from the option list); Ipopt is not in the
// reference tree, so what is built here is a restatement of its published algorithm (Waechter & Biegler, Math. Program.
// 106, 2006: primal-dual barrier, fraction-to-the-boundary rule, filter line search, inertia correction) for a batch of
// independent instances of one transcription — the MPC sweep of BASELINE config 5 — with every iterate, multiplier,
// KKT matrix and factor resident in HBM.  Restated beyond the basic iteration: bound_relax_factor, the second-order correction,
// the restoration phase (paper section 3.3, with a Gauss-Newton model of the constraint curvature) and least-squares
// multipliers on leaving it, Ipopt's adaptive barrier update with the LOQO oracle (option "mu_strategy", default 1; 0 = monotone) and its
// gradient-based NLP scaling (option "nlp_scaling", off by default).  Not restated: the quality-function oracle, the watchdog.  See DESIGN.md §f-2.
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

// One dense-ish sub-problem of the factorisation in band + border storage: element (i, j), i >= j, of its lower triangle
// sits at koff + j * CS + (i < Nb ? i - j : b + 1 + i - Nb) of the instance's KKT storage; its right-hand side is the
// slice [roff, roff + Nt) of the instance's vector.
struct KktSubHost {
  int Nt, Nb, nb, b, CS;
  long long koff;
  int roff;
};

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
  std::vector<int> hg_ptr, hg_src, hg_dst;   // Hessian by storage slot: K[hg_dst[i]] += (sum of hess[hg_src[...]] in COO order) — duplicates
                                             // (I-part / E-part, LpHessian.cpp:598) are summed in a fixed order, not by racing atomics
  // ---- nested dissection by mesh interval (nd = 1) ------------------------------------------------------------------
  // The unknowns of a mesh interval — states of its nodes but the first, controls, slacks, defect / path multipliers —
  // couple only to each other, to the states at the interval's first node ("separator"), to the separator of the NEXT
  // interval (last column of the interval's D block, Core/RPMGenerator.cpp:150-165) and to the border.  Level 1: every
  // interval is a band + border sub-problem of its own (border = its two separators + the global border; the first node's
  // controls, slacks and multipliers are ordered last inside it, see rpm_ipm.cpp), factored up to
  // its corner by a workgroup of its own; the corners are Schur complements that add up (in a fixed order) into level 2:
  // the separators in time order (block tridiagonal, blocks of nx) + the global border, one more band + border problem.
  // A quasi-definite matrix has an LDL^T under every symmetric permutation (Vanderbei 1995), so this ordering needs no
  // pivoting either, and the signs of all the D's still give the inertia.
  int nd = 0;
  int Nt_alloc = 0;                   // length of the right-hand-side vector: Nt plus the level-1 border work spaces
  long long storage_nd = 0;           // doubles of KKT storage per instance (level-1 blocks, then level 2)
  std::vector<KktSubHost> subs;       // level-1 sub-problems, [level-2 groups,] then ONE entry for the last level
  // ---- optional third level (n_l2 > 0): the separator system of a long mesh is itself banded (block tridiagonal), so it is cut
  // again, at the matrix level: consecutive groups of l3_S of its positions, the trailing l3_w (= its half bandwidth) of every
  // group but the last form that group's separator.  Group interiors are level-2 sub-problems (border = the previous
  // group's separator, its own, the global border), factored up to their corners by a workgroup each; all group separators in
  // order + the global border are the last level.  Delta-III 4 x 64 x 16: one chain of 112 block columns -> 8 + 16.
  int n_l2 = 0, l3_S = 0, l3_w = 0, l3_G = 0, l3_Nb2 = 0, l3_base = 0;
  std::vector<int> l3_gb_ptr, l3_gb;     // per group: the global-border unknowns (numbered inside the border) that have rows in its block, ascending
  int n_cg_long = 0, n_cg2_long = 0;   // the corner-gather tables start with the destinations that have >= 32 sources (a wave each on the device)
  std::vector<int> cg2_ptr, cg2_src, cg2_dst, rg2_ptr, rg2_src, rg2_dst, rs2_dst, rs2_src;   // second gather / scatter stage (level 2 -> last level)
  std::vector<int> cg_ptr, cg_src;    // corner gather: level-2 storage offset cg_dst[i] += sum of K[cg_src[cg_ptr[i] .. cg_ptr[i+1])]
  std::vector<int> cg_dst;
  std::vector<int> rg_ptr, rg_src;    // right-hand-side gather: rhs[rg_dst[i]] += sum of rhs[rg_src[...]]  (level-1 border work spaces)
  std::vector<int> rg_dst;
  std::vector<int> rs_dst, rs_src;    // solution scatter: rhs[rs_dst[i]] = rhs[rs_src[i]]
  std::vector<int> gap_pos;           // every position of a level-1 border work space (zeroed before the forward sweep)
  std::vector<int> nd_ivl, nd_loc, nd_sep, nd_sep0, nd_nsep, nd_last, nd_nstate0, nd_sep_state;   // classification kept for ipm_plan_offset
  std::vector<int> nd_gb_ptr, nd_gb;     // per interval: the global-border unknowns (numbered inside the border) that have rows in its block, ascending
  int max_rows = 0;                   // largest 16 + b + nb over all sub-problems (rows of a block column)
  size_t max_factor_lds = 0;
  long long storage() const { return nd ? storage_nd : (long long)Nt * CS; }
  long long at(int pa, int pb) const {        // pa >= pb
    return (long long)pb * CS + (pa < Nb ? pa - pb : b + 1 + pa - Nb);
  }
};

// rpm_ipm.cpp (host only): ordering, band width and the scatter maps from the engine's layout
int build_ipm_plan(Engine& e, IpmPlan& p, std::string* why, int nested = 0);   // nested: 0 one band, 1 by mesh interval, third level when the separator system is long (engine option ipm_nested_group)
void ipm_plan_group_hessian(IpmPlan& p);   // hg_* from hes_dst
// storage offset of the entry between unknowns ua, uc ([0,n) x, [n,nv) slacks, [nv,nv+m) multipliers); -1 if the layout has no slot for it
long long ipm_plan_offset(const IpmPlan& p, int ua, int uc);

}  // namespace rpm

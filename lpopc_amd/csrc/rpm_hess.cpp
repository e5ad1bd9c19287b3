// rpm_hess.cpp — exact-Hessian mode, host side: given the per-phase dependency patterns (probed on the
// device by ensure_hessian), build the lower-triangular COO structure and the tables the Hessian kernels
// index with.  Order and counts follow LpHessianCalculator::GetPhaseHessianSparsity (Core/LpHessian.cpp:601-876),
// GetLinkHessianSparsity (:2369-2508) and GetHessianSparsity (:2510-2600) for nq = 0; duplicates between the
// collocation part (I) and the endpoint part (E) are intentional (Ipopt sums them).
#include "rpm_engine.hpp"

namespace rpm {

void build_hessian_tables(Engine& e) {
  e.hess_pairs.clear();
  e.hess_phases.assign(e.P, HessPhaseDev{});
  e.hess_ends.clear();
  e.hess_links.clear();
  e.hes_i.clear();
  e.hes_j.clear();
  int v = 0, tmp = 0;
  for (int ip = 0; ip < e.P; ++ip) {
    const PhaseHost& p = e.ph[ip];
    const int nx = p.nx, nu = p.nu, nc = p.nc, nv = nx + nu, nout = nx + nc, N = p.N, sh = p.var0;
    const std::vector<int>& dep = e.hess_dep[ip];
    // H = dep' * dep with unit diagonal (LpHessian.cpp:899-903)
    std::vector<int> H(size_t(nv) * nv, 0);
    int nnzH = 0;
    for (int a = 0; a < nv; ++a)
      for (int b = 0; b < nv; ++b) {
        int acc = 0;
        for (int r = 0; r < nout; ++r) acc += dep[r + size_t(a) * nout] * dep[r + size_t(b) * nout];
        if (a == b) acc = 1;
        H[a + size_t(b) * nv] = acc;
        if (acc) ++nnzH;
      }
    HessPhaseDev& q = e.hess_phases[ip];
    q.pair0 = int(e.hess_pairs.size());
    q.v0 = v;
    q.nI = N * ((nnzH - nx - nu) / 2 + nx + nu) + 2 * (nx + nu) * N + 3;
    q.nE = (2 * nx) * (2 * nx - 1) / 2 + 2 * nx + 4 * nx + 3;
    q.end0 = int(e.hess_ends.size());
    q.tt_tmp = tmp;
    tmp += 3 * N;
    // pair records in the fixed role order a*(a+1)/2 + b over [x.., u.., t]
    const int NV = nv + 1;
    std::vector<HessPairDev> pairs(size_t(NV) * (NV + 1) / 2);
    for (int a = 0; a < NV; ++a)
      for (int b = 0; b <= a; ++b) pairs[size_t(a) * (a + 1) / 2 + b] = HessPairDev{a, b, 0, 0, 0};
    auto pair_of = [&](int a, int b) -> HessPairDev& { return pairs[size_t(a) * (a + 1) / 2 + b]; };
    std::vector<int> I_i(q.nI), I_j(q.nI), E_i(q.nE), E_j(q.nE);
    int sI = 0, sE = 0;
    auto block = [&](int rowstart, int colstart) {
      for (int k = 0; k < N; ++k) { I_i[sI] = sh + rowstart + k; I_j[sI++] = sh + colstart + k; }
    };
    auto rowblock = [&](int row, int colstart) {
      for (int k = 0; k < N; ++k) { I_i[sI] = sh + row; I_j[sI++] = sh + colstart + k; }
    };
    auto end_entry = [&](int a, int b, int da, int db, int row, int col) {
      e.hess_ends.push_back(HessEndDev{ip, a, b, da, db, q.nI + sE});
      E_i[sE] = sh + row;
      E_j[sE++] = sh + col;
    };
    for (int i = 0; i < nx; ++i)
      for (int j = 0; j <= i; ++j) {
        const int rs = i * (N + 1), cs = j * (N + 1);
        if (H[i + size_t(j) * nv]) {
          pair_of(i, j).kind = 1;
          pair_of(i, j).dst0 = sI;
          block(rs, cs);
        }
        end_entry(i, j, i, j, rs, cs);                         // x0_i, x0_j
        if (i != j) end_entry(i, nx + j, i, nx + i, rs, cs + N);   // x0_i, xf_j  (den pertx0(i)*pertxf(i), :1588)
        end_entry(nx + i, j, nx + i, j, rs + N, cs);           // xf_i, x0_j
        end_entry(nx + i, nx + j, nx + i, nx + i, rs + N, cs + N); // xf_i, xf_j  (den pertxf(i)*pertxf(i), :1612)
      }
    const int rowshift = nx * (N + 1);
    for (int i = 0; i < nu; ++i) {
      const int rs = rowshift + i * N;
      for (int j = 0; j < nx; ++j)
        if (H[(i + nx) + size_t(j) * nv]) {
          pair_of(nx + i, j).kind = 1;
          pair_of(nx + i, j).dst0 = sI;
          block(rs, j * (N + 1));
        }
      for (int j = 0; j <= i; ++j)
        if (H[(i + nx) + size_t(j + nx) * nv]) {
          pair_of(nx + i, nx + j).kind = 1;
          pair_of(nx + i, nx + j).dst0 = sI;
          block(rs, rowshift + j * N);
        }
    }
    const int trow = nx * (N + 1) + nu * N, T0 = 2 * nx, TF = 2 * nx + 1;
    for (int r = 0; r < 2; ++r) {
      const int row = trow + r, tv = r == 0 ? T0 : TF;
      for (int i = 0; i < nx; ++i) {
        HessPairDev& pr = pair_of(nv, i);
        pr.kind = 2;
        (r == 0 ? pr.dst0 : pr.dst1) = sI;
        rowblock(row, i * (N + 1));
        end_entry(tv, i, tv, i, row, i * (N + 1));
        end_entry(tv, nx + i, tv, nx + i, row, i * (N + 1) + N);
      }
      for (int i = 0; i < nu; ++i) {
        HessPairDev& pr = pair_of(nv, nx + i);
        pr.kind = 2;
        (r == 0 ? pr.dst0 : pr.dst1) = sI;
        rowblock(row, nx * (N + 1) + i * N);
      }
      if (r == 0) {
        q.tt_dst[0] = sI;
        I_i[sI] = sh + row; I_j[sI++] = sh + trow;
        end_entry(T0, T0, T0, T0, row, trow);
      } else {
        q.tt_dst[1] = sI;
        I_i[sI] = sh + row; I_j[sI++] = sh + trow;
        end_entry(TF, T0, TF, T0, row, trow);
        q.tt_dst[2] = sI;
        I_i[sI] = sh + row; I_j[sI++] = sh + trow + 1;
        end_entry(TF, TF, TF, TF, row, trow + 1);
      }
    }
    pair_of(nv, nv).kind = 3;
    q.n_end = int(e.hess_ends.size()) - q.end0;
    e.hess_pairs.insert(e.hess_pairs.end(), pairs.begin(), pairs.end());
    e.hes_i.insert(e.hes_i.end(), I_i.begin(), I_i.end());
    e.hes_j.insert(e.hes_j.end(), I_j.begin(), I_j.end());
    e.hes_i.insert(e.hes_i.end(), E_i.begin(), E_i.end());
    e.hes_j.insert(e.hes_j.end(), E_j.begin(), E_j.end());
    v += q.nI + q.nE;
  }
  for (int ip = 0; ip < e.L; ++ip) {
    const LinkDev& l = e.links[ip];
    const PhaseHost& pl = e.ph[l.left];
    const PhaseHost& pr = e.ph[l.right];
    const int nxl = pl.nx, nxr = pr.nx;
    for (int i = 0; i < nxl; ++i)
      for (int j = 0; j <= i; ++j) {
        e.hess_links.push_back(HessLinkDev{ip, i, j, v++});
        e.hes_i.push_back(pl.var0 + (pl.N + 1) * (i + 1) - 1);
        e.hes_j.push_back(pl.var0 + (pl.N + 1) * (j + 1) - 1);
      }
    for (int i = 0; i < nxr; ++i) {
      const int row = pr.var0 + (pr.N + 1) * i;
      for (int j = 0; j < nxl; ++j) {
        e.hess_links.push_back(HessLinkDev{ip, j, nxl + i, v++});   // hLink_xfL_x0R(j,i): first xfL_j, then x0R_i
        e.hes_i.push_back(row);
        e.hes_j.push_back(pl.var0 + (pl.N + 1) * (j + 1) - 1);
      }
      for (int j = 0; j <= i; ++j) {
        e.hess_links.push_back(HessLinkDev{ip, nxl + i, nxl + j, v++});
        e.hes_i.push_back(row);
        e.hes_j.push_back(pr.var0 + (pl.N + 1) * j);   // the reference uses nnodesLeft here (:2463)
      }
    }
  }
  e.nnz_h = v;
  e.hess_tmp_len = tmp;
  e.hess_ready = true;
}

}  // namespace rpm

// rpm_shard.cpp — interval sharding across the GPUs of a node.  Rank r owns a
// contiguous run of each phase's tiles; its share of g / values is a list of contiguous runs
// (one per output row-block resp. Jacobian block).  There is no reference counterpart: lpopc is
// single-process.  Host only.
#include "rpm_engine.hpp"

namespace rpm {

// node range [ka,kb) of phase `ip` owned by `rank`
static void owned_nodes(const Engine& e, int ip, int rank, int* ka, int* kb) {
  const int t0 = e.phd[ip].tile0, nt = e.phd[ip].ntiles, N = e.ph[ip].N;
  *ka = *kb = 0;
  bool any = false;
  for (int t = 0; t < nt; ++t) {
    const TileDev& tl = e.tiles[t0 + t];
    const int owner = int((long long)tl.k0 * e.shard_world / N);
    if (owner != rank) continue;
    if (!any) { *ka = tl.k0; any = true; }
    *kb = tl.k0 + tl.cnt;
  }
}

std::vector<rpm_segment> shard_segments(const Engine& e, int which, int rank, int* packed_len) {
  std::vector<rpm_segment> out;
  int pos = 0;
  auto add = [&](int off, int len) {
    if (len <= 0) return;
    out.push_back(rpm_segment{off, len, pos});
    pos += len;
  };
  for (int ip = 0; ip < e.P; ++ip) {
    const PhaseDev& q = e.phd[ip];
    int ka, kb;
    owned_nodes(e, ip, rank, &ka, &kb);
    const int NO = q.nx + q.nc, NB = q.nx + q.nu + 2 + q.nq;
    if (which == 0) {
      for (int o = 0; o < NO; ++o) add(q.g0 + o * q.N + ka, kb - ka);
      if (rank == 0) add(q.g0 + NO * q.N, q.ne);
    } else {
      for (int b = 0; b < NO * NB; ++b) add(q.v_nl0 + b * q.N + ka, kb - ka);
      if (rank == 0) add(q.v_evt0, q.ne * (2 * q.nx + 2 + q.nq));
    }
  }
  if (rank == 0) {
    for (int i = 0; i < e.L; ++i) {
      const LinkDev& l = e.links[i];
      if (which == 0) add(l.g0, l.nlink);
      else add(l.v0, l.nlink * (e.ph[l.left].nx + e.ph[l.left].nq + e.ph[l.right].nx + e.ph[l.right].nq));
    }
    if (which == 0) add(e.m_nl, e.P + e.L);
    else add(e.nnz_nl, e.nnz_lin);
  }
  if (which == 1) {
    // constant block: each tile copies a slice of its phase's Doffdiag list into every state's copy
    for (int ip = 0; ip < e.P; ++ip) {
      const PhaseDev& q = e.phd[ip];
      int q0 = -1, q1 = -1;
      for (int t = 0; t < q.ntiles; ++t) {
        const TileDev& tl = e.tiles[q.tile0 + t];
        if (int((long long)tl.k0 * e.shard_world / q.N) != rank) continue;
        if (q0 < 0) q0 = tl.c_src0 - q.doff_base;
        q1 = tl.c_src0 - q.doff_base + tl.c_cnt;
      }
      if (q0 < 0) continue;
      for (int i = 0; i < q.nx; ++i) add(e.nnz_nl + e.nnz_lin + q.const_cum + i * q.off_nnz + q0, q1 - q0);
    }
  }
  if (packed_len) *packed_len = pos;
  return out;
}

}  // namespace rpm

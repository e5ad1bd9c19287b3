// rpm_peer.hip — exchange of the interval-sharded results between the ranks of a node (north_star:
// "RCCL all-gather of the assembled constraint/Jacobian segments over xGMI").  The reference has no counterpart (lpopc is
// one process).  A rank's share of g and of the Jacobian values is a list of contiguous runs (rpm_shard.cpp).  For ONE
// collective per step the runs of BOTH vectors and of ALL instances of the rank are packed into one slot
//     slot(rank) = for each instance b:  [ g runs (plen_g[rank]) | values runs (plen_v[rank]) ]
// of a [world][slot_len] buffer: one pack kernel, one in-place all-gather over that buffer (RCCL, captured in the step's
// hipGraph by the caller), one unpack kernel that scatters the other ranks' slots into TNLP order.  No reductions: the
// result is bit-identical to the single-GPU vectors.
#include "rpm_device_internal.hpp"

namespace rpm {

struct XCopy {
  long long src, dst;     // offsets (doubles) of instance 0
  long long isrc, idst;   // added per instance
  int len;
  int which;              // 0: the g array, 1: the values array (the other side is the packed buffer)
};

template <bool PACK>
__global__ __launch_bounds__(256) void rpm_xcopy_kernel(const XCopy* __restrict__ tab, const double* __restrict__ a_g,
                                                        const double* __restrict__ a_v, double* __restrict__ o_g,
                                                        double* __restrict__ o_v, const double* __restrict__ packed_in,
                                                        double* __restrict__ packed_out) {
  const XCopy c = tab[blockIdx.x];
  const long long b = blockIdx.z;
  const double* src;
  double* dst;
  if (PACK) {
    src = (c.which ? a_v : a_g) + c.src + b * c.isrc;
    dst = packed_out + c.dst + b * c.idst;
  } else {
    src = packed_in + c.src + b * c.isrc;
    dst = (c.which ? o_v : o_g) + c.dst + b * c.idst;
  }
  for (int i = blockIdx.y * 256 + threadIdx.x; i < c.len; i += gridDim.y * 256) dst[i] = src[i];
}

struct Exchange {
  XCopy* d_pack = nullptr;
  XCopy* d_unpack = nullptr;        // every rank's runs
  XCopy* d_unpack_others = nullptr; // all but this rank's (its own results are in place already)
  int n_pack = 0, n_unpack = 0, n_unpack_others = 0;
  long long slot = 0;               // doubles per rank in the gathered buffer
  long long sg = 0, sv = 0;         // instance strides the tables were built for
};

long long shard_slot_len(const Engine& e) {
  long long most = 0;
  for (int r = 0; r < e.shard_world; ++r) {
    int pg = 0, pv = 0;
    (void)shard_segments(e, 0, r, &pg);
    (void)shard_segments(e, 1, r, &pv);
    most = std::max<long long>(most, (long long)pg + pv);
  }
  const long long slot = most * e.n_instances;
  return (slot + 15) / 16 * 16;   // whole 128-byte lines per rank
}

void exchange_destroy(Device* d) {
  if (!d->exchange) return;
  Exchange* x = static_cast<Exchange*>(d->exchange);
  for (void* p : {(void*)x->d_pack, (void*)x->d_unpack, (void*)x->d_unpack_others})
    if (p) (void)hipFree(p);
  delete x;
  d->exchange = nullptr;
}

static int exchange_setup(Engine& e) {
  Device& d = *e.dev;
  Exchange* x = static_cast<Exchange*>(d.exchange);
  const long long sg = e.stride_g(), sv = e.stride_values();
  if (x && x->sg == sg && x->sv == sv) return RPM_OK;
  exchange_destroy(&d);
  x = new Exchange();
  d.exchange = x;
  x->sg = sg;
  x->sv = sv;
  x->slot = shard_slot_len(e);
  std::vector<XCopy> pack, unpack, others;
  for (int r = 0; r < e.shard_world; ++r) {
    int pg = 0, pv = 0;
    const std::vector<rpm_segment> seg_g = shard_segments(e, 0, r, &pg), seg_v = shard_segments(e, 1, r, &pv);
    const long long per_inst = (long long)pg + pv;   // one instance's share inside rank r's slot
    for (int which = 0; which < 2; ++which)
      for (const rpm_segment& s : (which ? seg_v : seg_g)) {
        const long long ppos = (which ? pg : 0) + s.pos;           // inside one instance's share
        const long long istride = which ? sv : sg;
        if (r == e.shard_rank) pack.push_back(XCopy{s.off, ppos, istride, per_inst, s.len, which});
        const XCopy u{r * x->slot + ppos, s.off, per_inst, istride, s.len, which};
        unpack.push_back(u);
        if (r != e.shard_rank) others.push_back(u);
      }
  }
  auto up = [&](const std::vector<XCopy>& v, XCopy** dst, int* n) -> int {
    *n = int(v.size());
    HIP_TRY(e, hipMalloc(reinterpret_cast<void**>(dst), (v.size() ? v.size() : 1) * sizeof(XCopy)));
    if (!v.empty()) HIP_TRY(e, hipMemcpy(*dst, v.data(), v.size() * sizeof(XCopy), hipMemcpyHostToDevice));
    return RPM_OK;
  };
  int rc;
  if ((rc = up(pack, &x->d_pack, &x->n_pack))) return rc;
  if ((rc = up(unpack, &x->d_unpack, &x->n_unpack))) return rc;
  if ((rc = up(others, &x->d_unpack_others, &x->n_unpack_others))) return rc;
  return RPM_OK;
}

int dev_shard_pack_all(Engine& e, const double* d_g, const double* d_values, double* d_slot, void* stream) {
  if (!e.dev) {
    int rc = device_init(e, 0);
    if (rc) return rc;
  }
  int rc = exchange_setup(e);
  if (rc) return rc;
  Exchange& x = *static_cast<Exchange*>(e.dev->exchange);
  if (!x.n_pack) return RPM_OK;
  hipLaunchKernelGGL(rpm_xcopy_kernel<true>, dim3(unsigned(x.n_pack), 2, unsigned(e.n_instances)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), x.d_pack, d_g, d_values, nullptr, nullptr, nullptr, d_slot);
  HIP_TRY(e, hipGetLastError());
  return RPM_OK;
}

int dev_shard_unpack_all(Engine& e, const double* d_gathered, double* d_g, double* d_values, int skip_own, void* stream) {
  if (!e.dev) {
    int rc = device_init(e, 0);
    if (rc) return rc;
  }
  int rc = exchange_setup(e);
  if (rc) return rc;
  Exchange& x = *static_cast<Exchange*>(e.dev->exchange);
  const int n = skip_own ? x.n_unpack_others : x.n_unpack;
  if (!n) return RPM_OK;
  hipLaunchKernelGGL(rpm_xcopy_kernel<false>, dim3(unsigned(n), 2, unsigned(e.n_instances)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), skip_own ? x.d_unpack_others : x.d_unpack, nullptr, nullptr, d_g, d_values,
                     d_gathered, nullptr);
  HIP_TRY(e, hipGetLastError());
  return RPM_OK;
}

}  // namespace rpm

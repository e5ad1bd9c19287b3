// rpm_engine.hpp — engine state shared by the host set-up code (rpm_setup.cpp), the C ABI
// (rpm_abi.cpp) and the HIP side (rpm_device.hip).  Plain C++17, no HIP types here so the
// set-up half builds and runs without a GPU.
//
// Reference counterparts (paths relative to /root/reference/Lpopc/src):
//   PhaseHost  <- struct ps + struct indices, Core/LpCalculateData.hpp:29-41
//   Engine     <- LpCalculateData, Core/LpCalculateData.hpp:43-111 (the per-mesh blackboard)
#pragma once
#include <cstddef>
#include <string>
#include <utility>
#include <vector>

#include "../../include/rpm_hip.h"

namespace rpm {

// ---- device-visible POD tables (copied to HBM verbatim by rpm_device_init) ---------------
struct PhaseDev {
  int N, nx, nu, nc, ne;
  int nq;            // static parameters: x[x_t0 + 2 + j]
  int phase_num;     // 1-based, what the reference hands the user callbacks (LpNLPWrapper.cpp:107)
  int x_state0;      // index of X(0,0) in x          (phase_indices[i]->state[0]-1)
  int x_control0;    // index of U(0,0) in x
  int x_t0;          // index of t0 in x; tf = x_t0+1
  int g0;            // first constraint row of the phase
  int v_nl0;         // offset of the phase's NL block in `values`
  int v_evt0;        // offset of the phase's event entries in `values`
  int node0;         // offset of the phase in the per-node tables
  int doff_base;     // offset of the phase's Doffdiag values in doff_vals
  int off_nnz;       // number of Doffdiag entries of the phase
  int const_cum;     // offset of the phase inside the CONST block of `values`
  int tile0, ntiles; // tiles of this phase (all ranks)
};

// One workgroup's share of a phase: a run of consecutive collocation nodes.
// Everything a workgroup needs is in this one record (one scalar load burst, no dependent
// second lookup of the phase table on the critical path).
struct TileDev {
  int phase;
  int k0, cnt;       // nodes [k0, k0+cnt) of the phase
  int span0, span_len;  // rows of the (N+1) x nx state matrix the tile's D rows touch
  int drow0, drow_len;  // the tile's D rows inside dvals (row-major per node, contiguous)
  // copied from the phase record
  int N, phase_num, x_state0, x_control0, x_t0, g0, v_nl0, node0;
  // this tile's share of the phase's constant Doffdiag block: source doff_vals[c_src0 .. +c_cnt),
  // written to values[c_dst0 + i*c_stride + q], i = 0..c_copies-1 (one copy per state)
  int c_src0, c_cnt, c_dst0, c_stride, c_copies;
};

// ---- exact-Hessian mode (hessian-approximation=exact), LpHessian.cpp ------------------------------
// One record per perturbation pair (a >= b over the node variables [x.., u.., t]) of a phase.
struct HessPairDev {
  int a, b;
  int kind;        // 0: not stored (pattern H(a,b) = 0), 1: N-long block, 2: (t, v) pair -> t0 and tf row blocks, 3: (t, t)
  int dst0, dst1;  // offsets inside the phase's I-part (kind 1: dst0; kind 2: t0 block, tf block)
};
struct HessPhaseDev {
  int pair0;             // first HessPairDev of the phase
  int v0;                // offset of the phase's entries in the Hessian values
  int nI, nE;
  int tt_dst[3];         // I-part offsets of the t0t0, tft0, tftf scalars
  int end0, n_end;       // endpoint entries (HessEndDev) of the phase
  int tt_tmp;            // offset of the phase's per-node tt terms in the scratch array (3 x N)
};
// One E-part (events + Mayer) entry: second difference w.r.t. endpoint variables a then b of
// W = [x0.., xf.., t0, tf]; the denominator is pert(da)*pert(db) exactly as the reference writes it
// (including its pertxf(istate) quirk, LpHessian.cpp:1588,1612,1826,1850).
struct HessEndDev { int phase, a, b, da, db, dst; };
// One linkage entry: variables a then b of [xf_left.., x0_right..]
struct HessLinkDev { int pair, a, b, dst; };

// endpoint work items, one workgroup each: 0 = linear rows, 1 = events of phase idx, 2 = linkage pair idx
struct TaskDev { int type, idx; };

struct NodeDev {     // per collocation node
  int drow_off;      // offset of the node's D row in dvals
  int dcol0;         // first state-matrix row (= column of D) of that row
  int dlen;          // N_k + 1
  int interval;      // mesh interval index inside the phase
};

struct LinkDev {
  int left, right;   // 0-based phases
  int nlink;
  int g0;            // first constraint row
  int v0;            // offset of the pair's entries in `values`
};

// ---- mesh-error estimate (row f-3): per mesh interval, offsets into the phase's interpolation / integration tables
struct MeshIvDev {
  int n;        // collocation nodes of the interval on the current mesh (the finer mesh has n + 1)
  int istart;   // first node of the interval on the current mesh
  int r0;       // first row of the interval on the finer mesh
  int q0;       // offset of its n + 1 rows in ttem / Ss / Sc / hit_s / hit_c
  int hs, hc;   // offsets of its state ((n+1) x (n+1)) and control ((n+1) x n) interpolation rows, column-major
  int a;        // offset of its (n+1) x (n+1) integration matrix, column-major
};
struct MeshErrTables {
  std::vector<MeshIvDev> iv;
  std::vector<double> ttem, Hs, Ss, Hc, Sc, A;
  std::vector<int> hit_s, hit_c;
  int fine_nodes = 0, rows = 0;   // sum(n_k + 1) and that + 1
};

// ---- host-side phase tables ----------------------------------------------------------
struct PhaseHost {
  int nx = 0, nu = 0, nq = 0, nc = 0, ne = 0, K = 0, N = 0;
  std::vector<double> mesh;
  std::vector<int> nk;
  std::vector<double> points, weights;            // ps.Points / ps.Weights
  std::vector<int> d_i, d_j;                       // ps.D triplets, reference order
  std::vector<double> d_v;
  std::vector<double> diag_v;                      // ps.Diag values (N)
  std::vector<int> off_i, off_j;                   // ps.Doffdiag triplets
  std::vector<double> off_v;
  std::vector<double> drows;                       // D rows, row-major per node (device layout)
  std::vector<NodeDev> nodes;
  int var0 = 0, con0 = 0, nvar = 0, ncon = 0;
};

// page-locked staging slots of an engine (Device::Stage): caller arrays that are not registered go through these
enum { STAGE_X = 0, STAGE_G, STAGE_V, STAGE_GRAD, STAGE_LAMBDA, STAGE_HESS, STAGE_SLOTS };

struct Device;  // HIP-side state, defined in rpm_device.hip

struct Engine {
  // description
  int problem_id = 0, P = 0, L = 0;
  std::vector<double> consts;
  std::vector<double> inst_consts;   // n_instances x consts.size() once rpm_set_instance_constants was used, else empty (shared)
  double fd_tol = 1e-6;
  int first_derive = 0, hessian_mode = 0;
  int n_instances = 1;
  int shard_mode = 0, shard_rank = 0, shard_world = 1;
  // per-mesh data
  std::vector<PhaseHost> ph;
  std::vector<LinkDev> links;
  std::vector<std::vector<double>> link_min, link_max;
  int n = 0, m_nl = 0, m = 0;
  int nnz_nl = 0, nnz_lin = 0, nnz_const = 0, nnz_jac = 0, nnz_h = 0;
  std::vector<double> xl, xu, gl, gu, guess;
  std::vector<int> alin_i, alin_j;                // A_lin triplets (LpBoundsChecker.cpp:265-346)
  std::vector<double> alin_v;
  std::vector<int> jac_i, jac_j;                  // cached structure (NLPWrapper::GetConsSparsity)
  std::vector<int> hes_i, hes_j;
  bool hess_ready = false;                         // dependency probe done, structure + tables built
  std::vector<std::vector<int>> hess_dep;          // per phase (nx+nc) x (nx+nu), column-major 0/1
  std::vector<HessPairDev> hess_pairs;
  std::vector<HessPhaseDev> hess_phases;
  std::vector<HessEndDev> hess_ends;
  std::vector<HessLinkDev> hess_links;
  int hess_tmp_len = 0;
  // device tables (host copies)
  std::vector<PhaseDev> phd;
  std::vector<TileDev> tiles;                      // all tiles, phase-major
  std::vector<int> my_tiles;                       // indices of the tiles this rank computes
  std::vector<TaskDev> tasks;                      // endpoint work items (rank 0 of a sharded run)
  std::vector<NodeDev> nodes;                      // all phases concatenated
  std::vector<double> points, weights, diag, dvals, doff_vals;
  int tile_nodes = 16;
  bool role_looped = false;   // T = 64 nodes x 4 role groups, roles walked sequentially (large grids)
  int max_span = 0, max_drow = 0;
  // options
  int opt_fuse_pair = 1, opt_dx_mode = 0, opt_tile_nodes = 0, opt_check_finite = 1, opt_const_once = 0;
  int opt_role_loop = -1;        // -1 automatic, 0 never, 1 always (when tile_nodes is automatic)
  int opt_instance_align = 1;    // device-resident batched calls: every instance's g / values array starts on a multiple of this many doubles
  long long stride_g() const { return (long long)(m + opt_instance_align - 1) / opt_instance_align * opt_instance_align; }
  long long stride_values() const { return (long long)(nnz_jac + opt_instance_align - 1) / opt_instance_align * opt_instance_align; }
  int opt_stage_roles = -1;      // pipelined kernel: functors with a staged dae (problems.hpp has_stage) recompute per role only what the perturbed variable enters; -1: when the constant block is skipped
  int opt_pipeline = -1;         // rpm_tile_pl_kernel: -1 automatic (>= 2 tiles per resident workgroup), 0 never, 1 whenever the mesh fits
  int jac_nonfinite = -1;                 // verdict on the cached Jacobian of the fused pair launch (-1: not checked)
  const double* const_filled = nullptr;   // host `values` buffer whose LIN/CONST tail this engine wrote last (const_once)
  int opt_persistent_values = 0; // device-resident Jacobian calls: skip the constant block of a `values` array this engine filled before (rpm_hip.h)
  int opt_pin_host = 0;          // 1: page-lock the caller's x / g / values arrays through the process-wide registry (librpm_pin.so); opt-in
  std::vector<std::pair<const void*, size_t>> pin_refused;   // arrays the registry refused this engine (not asked for again until "pin_host" is set again)
  std::string pin_note;          // why the last registration this engine asked for was refused (shown by rpm_last_error after the reason of a failed call)
  int opt_zero_copy = 1;         // host-pointer path: the kernel reads x from / stores g into the caller's page-locked arrays
  int opt_delta_values = 0;      // host-pointer path: deliver only the runs of `values` that changed since the last delivery into the same array
  int last_delta_total = 0;      // runs of `values` this engine owns (rpm_get_option "delta_total_runs")
  int opt_ipm_nested_group = 0;  // rpm_ipm_create: positions per group when the separator system of the nested dissection is cut again; 0 = automatic (only when it is >= 512 long)
  int opt_ipm_local_border = 1;  // rpm_ipm_create: an interval block carries only the rows of the global border that its interior has entries with (0: all of them)
  int opt_ipm_nested = -1;       // rpm_ipm_create: nested dissection of the KKT matrix over the mesh intervals (rpm_ipm.hpp): -1 when the structure allows, 0 never, 1 must
  int ipm_attached = 0;          // rpm_ipm solvers built on this engine: they size their buffers from stride_g/values
  // solution kept by finalize_solution (LpopcIpopt.cpp:237-243)
  std::vector<double> sol_x, sol_lambda;
  double sol_obj = 0.0;
  bool has_solution = false;
  std::vector<MeshErrTables> mesh_err;            // built on first use, per phase
  // state
  std::string err;
  std::string err_report;   // what rpm_last_error hands out when the registry has something to add
  Device* dev = nullptr;
};

// rpm_setup.cpp
int setup_engine(Engine& e, const rpm_problem_desc* d);  // returns RPM_* code, message in e.err
void build_tiles(Engine& e, int tile_nodes);
void lgr_points(int n, std::vector<double>& x, std::vector<double>& w);
void colloc_d(const std::vector<double>& pts, std::vector<double>& D);  // (M-1) x M, column-major

// rpm_hess.cpp: Hessian structure and device tables from the dependency patterns (host only)
void build_hessian_tables(Engine& e);
// rpm_device.hip: runs the NaN-propagation dependency probe on the device, then build_hessian_tables
int ensure_hessian(Engine& e);
int dev_eval_h(Engine& e, const double* d_x, double obj_factor, const double* d_lambda, double* d_values, void* stream);

int dev_nlp2op(Engine& e, int phase, const double* x, const double* lambda, double* time, double* state, double* control,
               double* costate, double* pathmult, double* hamiltonian, double* mayer_cost, double* lagrange_cost);

// rpm_hpliu.cpp: LiuHpMeshRefineAlg's state across meshes (mesh_history_, state_history_, mesh_points_history_)
struct HpLiu {
  struct Mesh { std::vector<double> mesh; std::vector<int> nodes; std::vector<double> e_k; };
  struct Solution { int rows = 0; std::vector<double> v; };
  double tol = 1e-6, R = 1.2;
  int Nmax = 16, mesh_index = 0;
  std::vector<std::vector<Mesh>> meshes;
  std::vector<std::vector<Solution>> states;
  std::vector<std::vector<std::vector<double>>> points_hist;
  int refine(const Engine& e, const double* x, const std::vector<std::vector<double>>& rel,
             std::vector<std::vector<double>>& new_mesh, std::vector<std::vector<int>>& new_nodes, bool* no_more,
             std::string* why);
  int smooth_enough(int ip, int first_row, int n, const std::vector<double>& tau, const double* state, int ld, int nx) const;
  bool exponent(int ip, double m0, double mf, int N, double e_k, double* q) const;
};
void lagrange_rows(const double* src, int m, const double* dst, int nq, double* H, double* S, int* hit);

// rpm_mesh.cpp: tables of the mesh-error estimate and the ph refinement decision (host); rpm_device.hip: the estimate
void build_mesh_err_tables(const PhaseHost& p, MeshErrTables& t);
bool ph_refine(const PhaseHost& p, const double* rel, double tol, int nmin, int nmax, std::vector<double>& mesh,
               std::vector<int>& nodes, std::vector<double>& interval_error);
int dev_solution_error(Engine& e, int phase, const double* x, double* rel_err);

// rpm_shard.cpp: rank's contiguous runs of g (which=0) or of the Jacobian values (which=1)
std::vector<rpm_segment> shard_segments(const Engine& e, int which, int rank, int* packed_len);
int dev_shard_copy(Engine& e, int which, bool pack, const double* src, int stride, double* dst, void* stream);
// rpm_peer.hip: ONE packed slot per rank for g + values of all instances (one collective per step)
long long shard_slot_len(const Engine& e);
int dev_shard_pack_all(Engine& e, const double* d_g, const double* d_values, double* d_slot, void* stream);
int dev_shard_unpack_all(Engine& e, const double* d_gathered, double* d_g, double* d_values, int skip_own, void* stream);

// problem registry (rpm_device.hip): static dimensions of a functor, for validation on the host
struct ProblemDims { int nx, nu, nc, ne_max, nlink_max, nconst; bool has_analytic; int nq; };
bool problem_dims(int problem_id, ProblemDims* out);

// rpm_device.hip
int device_init(Engine& e, int device_id);
void device_destroy(Engine& e);
// flags: bit0 = constraint vector g, bit1 = Jacobian values.  Pointers are device pointers.
int dev_eval_cons(Engine& e, const double* d_x, double* d_g, double* d_values, int flags, void* stream);
// objective (d_obj, one per instance) and, when d_grad != nullptr, its gradient
int dev_eval_obj(Engine& e, const double* d_x, double* d_obj, double* d_grad, void* stream, bool host_chk = false);
bool dev_cons_is_one_role(const Engine& e);  // the next constraint launch is rpm_tile_kernel (carries the fused NaN/Inf check)
// rpm_host_path.hip: the host-pointer TNLP path (x, g, values are the caller's arrays)
int host_eval_f(Engine& e, const double* x, int new_x, double* obj);
int host_eval_grad_f(Engine& e, const double* x, int new_x, double* grad);
int host_eval_g(Engine& e, const double* x, int new_x, double* g);
void host_new_x(Engine& e);   // a callback received new_x = true: drop what the other callbacks cached
int host_eval_jac_values(Engine& e, const double* x, int new_x, double* values);
int host_eval_pair(Engine& e, const double* x, double* g, double* values);
// the constraint callbacks in two halves (rpm_host_path.hip): everything queued / synchronised, delivered and judged
struct ConsCall { bool want_g, want_v, cached, delta, launched_jac; };
int host_cons_begin(Engine& e, const double* x, int new_x, double* g, double* values, ConsCall* c);
int host_cons_end(Engine& e, double* g, double* values, const ConsCall& c, const char* who);
int host_delta_sent_runs(Engine& e, int* sent);   // runs delivered since the previous query
int dev_pipeline_active(const Engine& e);   // 1 when the next constraint launch uses rpm_tile_pl_kernel
int dev_upload_x(Engine& e, const double* x);
int dev_upload(Engine& e, double* dev, const double* host, size_t count, int slot);    // slot: STAGE_* (rpm_device_internal.hpp)
int dev_download(Engine& e, double* host, const double* dev, size_t count, int slot);
int dev_sync(Engine& e);
int dev_update_instance_constants(Engine& e);   // (re)uploads e.inst_consts and points the kernels at it
double* dev_buf(Engine& e, int which);  // 0 x, 1 g, 2 values, 3 grad, 4 obj, 5 lambda, 6 hess
bool& dev_cache_valid(Engine& e);
void dev_forget_persistent(Engine& e);   // no device `values` array counts as filled any more (option "persistent_values")
void* dev_stream(Engine& e);
int dev_nonfinite(Engine& e, const double* dev, size_t count);   // 1 if a NaN/Inf is present, checked on the device
int dev_nonfinite_enqueue(Engine& e, const double* a, size_t na, const double* b, size_t nb);   // asynchronous form, flag words 0 / 1
int dev_flag_value(Engine& e, int slot);
long dev_pin_counter(int which);  // process-wide counters of librpm_pin.so (RPM_PIN_*, rpm_pin.h)
int dev_pin_held(const Engine& e);                                 // registrations this engine holds
void* dev_pin_host(Engine& e, const void* ptr, size_t bytes);      // "pin_host": the array's device alias once its pages are registered, else nullptr
void dev_pin_release_all(Engine& e);   // let go of every registration this engine holds (rpm_set_option "pin_host" 0, rpm_destroy)

}  // namespace rpm

// the opaque handle of include/rpm_hip.h
struct rpm_engine {
  rpm::Engine e;
};

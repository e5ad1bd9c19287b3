"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs.  Tolerances (north_star: "within a stated floating-point tolerance"):
  eval_g, eval_f             |d| <= 1e-12 * max(1,|ref|)   (libm differences exp/pow/acos only)
  eval_jac_g finite diff.    |d| <= 1e-8  * max(1,|ref|)   (a 1-ulp difference in f divided by h ~ 1e-6)
  eval_jac_g analytic, constant and linear entries          bit-exact / 1e-12
"""
import numpy as np
import pytest

from lpopc_amd import problems
from lpopc_amd.engine import NLPEngine
from lpopc_amd.problem import Options

pytestmark = pytest.mark.gpu

G_TOL, JFD_TOL, JAN_TOL = 1e-12, 1e-8, 1e-12


def rel_err(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))) if a.size else 0.0


def oracle_for(prob, opts=None):
    from oracle.oracle import Oracle
    return Oracle(prob, opts)


CASES = [
    ("brachistochrone", lambda: problems.brachistochrone(1, 10), "perturb"),
    ("bryson_denham_default", lambda: problems.bryson_denham(), "perturb"),
    ("launch_default_1x20", lambda: problems.launch(), "perturb"),
    ("launch_ragged", lambda: _launch_ragged(), "perturb"),
    ("climb_16x16", lambda: problems.min_time_climb(16, 16), "perturb"),
    ("hypersensitive_hp", lambda: problems.config("hypersensitive"), "uniform"),
    ("quadrotor_8x8", lambda: problems.quadrotor(8, 8), "perturb"),
    ("launch_metric_64x16", lambda: problems.launch(64, 16), "perturb"),
]


def _launch_ragged():
    """Delta-III on ragged hp meshes: unequal widths, N_k from 2 to 23 (wider than a 16-node tile)."""
    p = problems.launch()
    meshes = [([-1, -0.6, 0.1, 1], [5, 23, 2]), ([-1, 0.5, 1], [16, 17]), ([-1, 1], [33]),
              ([-1, -0.9, -0.5, 0.0, 0.25, 1], [3, 4, 7, 12, 16])]
    for i, (mesh, nodes) in enumerate(meshes):
        problems.set_mesh(p.GetPhase(i), mesh, nodes)
    return p


@pytest.mark.parametrize("name,make,mode", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("tile", [0, 16, 32, 64])
def test_eval_g_and_jac_fd(built, name, make, mode, tile):
    prob = make()
    eng = NLPEngine(prob, tile_nodes=tile, device=0)
    orc = oracle_for(prob)
    xl, xu, _, _ = eng.get_bounds_info()
    for seed in (3, 11):
        x = problems.seeded_iterate(eng.get_starting_point(), xl, xu, seed, mode)
        g = eng.eval_g(x, True)
        v = eng.eval_jac_g(x, False)      # served from the fused launch of eval_g
        g_ref, v_ref = orc.eval_g(x), orc.eval_jac_g(x)
        assert rel_err(g, g_ref) <= G_TOL
        assert rel_err(v, v_ref) <= JFD_TOL
        # the constant (Doffdiag) and linear blocks are copies: bit-exact
        assert np.array_equal(v[-_n_const(eng):], v_ref[-_n_const(eng):])
        # separate (unfused) kernels give the same numbers
        eng.set_option("fuse_pair", 0)
        assert np.array_equal(eng.eval_g(x, True), g)
        assert np.array_equal(eng.eval_jac_g(x, True), v)
        eng.set_option("fuse_pair", 1)
    eng.close()


def _n_const(eng):
    tot = 0
    for p in range(eng.n_phases):
        tot += eng.phase_tables(p)["doff_vals"].size * _nx(eng, p)
    return tot


def _nx(eng, p):
    return eng._desc.phases[p].nx


@pytest.mark.parametrize("name,make", [("hypersensitive", lambda: problems.config("hypersensitive")),
                                       ("brachistochrone", lambda: problems.brachistochrone(3, 7))])
def test_analytic_mode(built, name, make):
    prob = make()
    opts = Options()
    opts.SetStringValue("first-derive", "analytic")
    eng = NLPEngine(prob, opts, device=0)
    orc = oracle_for(prob, opts)
    xl, xu, _, _ = eng.get_bounds_info()
    x = problems.seeded_iterate(eng.get_starting_point(), xl, xu, 5, "uniform")
    assert rel_err(eng.eval_g(x), orc.eval_g(x)) <= G_TOL
    assert rel_err(eng.eval_jac_g(x), orc.eval_jac_g(x)) <= JAN_TOL
    assert abs(eng.eval_f(x) - orc.eval_f(x)) <= 1e-12 * max(1.0, abs(orc.eval_f(x)))
    assert rel_err(eng.eval_grad_f(x), orc.eval_grad_f(x)) <= 1e-12
    eng.close()


@pytest.mark.parametrize("name,make,mode", CASES[:7], ids=[c[0] for c in CASES[:7]])
def test_objective_and_gradient(built, name, make, mode):
    prob = make()
    eng = NLPEngine(prob, device=0)
    orc = oracle_for(prob)
    xl, xu, _, _ = eng.get_bounds_info()
    x = problems.seeded_iterate(eng.get_starting_point(), xl, xu, 7, mode)
    f, f_ref = eng.eval_f(x), orc.eval_f(x)
    assert abs(f - f_ref) <= 1e-12 * max(1.0, abs(f_ref))
    assert rel_err(eng.eval_grad_f(x), orc.eval_grad_f(x)) <= 1e-8
    eng.close()


def test_jacobian_matches_central_differences_of_eval_g(built):
    """Oracle-free check: J(x) dx ~ (g(x+e dx) - g(x-e dx)) / 2e along random directions."""
    import scipy.sparse as sp
    prob = problems.launch(3, 6)
    eng = NLPEngine(prob, device=0)
    xl, xu, _, _ = eng.get_bounds_info()
    x = problems.seeded_iterate(eng.get_starting_point(), xl, xu, 9)
    i, j = eng.eval_jac_g_structure()
    J = sp.coo_matrix((eng.eval_jac_g(x), (i, j)), shape=(eng.m, eng.n)).tocsr()
    rng = np.random.RandomState(0)
    for _ in range(3):
        dx = rng.uniform(-1, 1, eng.n)
        e = 1e-6
        fd = (eng.eval_g(x + e * dx) - eng.eval_g(x - e * dx)) / (2 * e)
        assert np.max(np.abs(J @ dx - fd)) <= 2e-4 * max(1.0, np.max(np.abs(fd)))  # forward-difference truncation O(h f")
    eng.close()


def test_batched_instances(built):
    """MPC sweep: B structurally identical instances in one launch == B single evaluations."""
    import torch
    B = 37
    prob = problems.quadrotor(8, 8)
    one = NLPEngine(prob, device=0)
    many = NLPEngine(prob, n_instances=B, device=0)
    xl, xu, _, _ = one.get_bounds_info()
    xs = np.stack([problems.seeded_iterate(one.get_starting_point(), xl, xu, 5 + b) for b in range(B)])
    dx = torch.from_numpy(xs).cuda()
    dg = torch.empty((B, one.m), dtype=torch.float64, device="cuda")
    dv = torch.empty((B, one.nnz_jac), dtype=torch.float64, device="cuda")
    many.eval_pair_dev(dx, dg, dv)
    torch.cuda.synchronize()
    for b in (0, 1, B // 2, B - 1):
        assert np.array_equal(dg[b].cpu().numpy(), one.eval_g(xs[b]))
        assert np.array_equal(dv[b].cpu().numpy(), one.eval_jac_g(xs[b], False))
    f = many.eval_f(xs)
    assert f.shape == (B,) and abs(f[3] - one.eval_f(xs[3])) == 0.0
    one.close()
    many.close()


def test_full_size_properties(built):
    """Metric config (4 x 64 x 16): size-independent properties — determinism, linearity of the
    D.X part in the states, structure/value consistency."""
    prob = problems.launch(64, 16)
    eng = NLPEngine(prob, device=0)
    assert (eng.n, eng.m, eng.nnz_jac) == (40996, 32801, 852356)
    xl, xu, _, _ = eng.get_bounds_info()
    x = problems.seeded_iterate(eng.get_starting_point(), xl, xu, 3)
    g1, v1 = eng.eval_g(x), eng.eval_jac_g(x, False)
    g2, v2 = eng.eval_g(x), eng.eval_jac_g(x, False)
    assert np.array_equal(g1, g2) and np.array_equal(v1, v2)          # deterministic
    assert np.all(np.isfinite(g1)) and np.all(np.isfinite(v1))
    i, j = eng.eval_jac_g_structure()
    assert len(set(zip(i.tolist(), j.tolist()))) == eng.nnz_jac          # no duplicate (i,j) entries
    eng.close()


def test_nonfinite_is_reported(built):
    prob = problems.launch(2, 4)
    eng = NLPEngine(prob, device=0)
    x = eng.get_starting_point()
    x[0] = np.nan
    with pytest.raises(Exception) as ei:
        eng.eval_g(x)
    assert "non-finite" in str(ei.value)
    eng.close()


def test_interval_sharding_on_one_gpu(built):
    """"Fake rank" mode: the shards of an 8-way interval sharding run one after another on one
    GPU; packing each rank's runs, concatenating as an all-gather would and unpacking reproduces the
    unsharded vectors bit for bit."""
    import torch
    prob = problems.launch(16, 8)
    world = 8
    ref = NLPEngine(prob, device=0)
    xl, xu, _, _ = ref.get_bounds_info()
    x = problems.seeded_iterate(ref.get_starting_point(), xl, xu, 3)
    g_ref, v_ref = ref.eval_g(x), ref.eval_jac_g(x, False)
    dx = torch.from_numpy(x).cuda()
    engs = [NLPEngine(prob, shard_mode=1, shard_rank=r, shard_world=world, device=0) for r in range(world)]
    stride = [max(engs[0].shard_segments(w, r)[1] for r in range(world)) for w in (0, 1)]
    gathered = [torch.zeros(world * stride[w], dtype=torch.float64, device="cuda") for w in (0, 1)]
    for r, e in enumerate(engs):
        dg = torch.full((e.m,), float("nan"), dtype=torch.float64, device="cuda")
        dv = torch.full((e.nnz_jac,), float("nan"), dtype=torch.float64, device="cuda")
        e.eval_pair_dev(dx, dg, dv)
        for w, full in ((0, dg), (1, dv)):
            e.shard_pack_dev(w, full, gathered[w][r * stride[w]:(r + 1) * stride[w]])
    out_g = torch.full((ref.m,), float("nan"), dtype=torch.float64, device="cuda")
    out_v = torch.full((ref.nnz_jac,), float("nan"), dtype=torch.float64, device="cuda")
    engs[3].shard_unpack_dev(0, gathered[0], stride[0], out_g)
    engs[3].shard_unpack_dev(1, gathered[1], stride[1], out_v)
    torch.cuda.synchronize()
    og, ov = out_g.cpu().numpy(), out_v.cpu().numpy()
    bad_g, bad_v = np.flatnonzero(~(og == g_ref)), np.flatnonzero(~(ov == v_ref))
    assert bad_g.size == 0, (bad_g[:10], bad_g.size)
    assert bad_v.size == 0, (bad_v[:10], bad_v.size, ref.nnz_jac, ov[bad_v[:6]], v_ref[bad_v[:6]], np.unique(bad_v // 128)[:40])
    for e in engs + [ref]:
        e.close()


@pytest.mark.parametrize("name,make,mode", [CASES[2], CASES[3], CASES[5], CASES[6], CASES[7]],
                         ids=[CASES[i][0] for i in (2, 3, 5, 6, 7)])
@pytest.mark.parametrize("tile", [16, 32, 64])
def test_mfma_dx_mode(built, name, make, mode, tile):
    """dx_mode=1: the tile's D.X on the FP64 matrix cores (v_mfma_f64_16x16x4_f64).  Same result as the
    scalar reference-order sum up to the summation order (<= 1e-12 relative here); the Jacobian is untouched."""
    prob = make()
    eng = NLPEngine(prob, tile_nodes=tile, device=0)
    orc = oracle_for(prob)
    xl, xu, _, _ = eng.get_bounds_info()
    x = problems.seeded_iterate(eng.get_starting_point(), xl, xu, 13, mode)
    g0, v0 = eng.eval_g(x), eng.eval_jac_g(x, False)
    eng.set_option("dx_mode", 1)
    g1, v1 = eng.eval_g(x), eng.eval_jac_g(x, False)
    assert rel_err(g1, orc.eval_g(x)) <= 5e-12
    assert rel_err(g1, g0) <= 5e-12   # summation order only; terms |d_j x_j| reach 1e3 on 1/64-wide intervals
    assert np.array_equal(v1, v0)
    eng.set_option("fuse_pair", 0)
    assert np.array_equal(eng.eval_g(x), g1)
    eng.close()


@pytest.mark.parametrize("name,make,mode,B", [(CASES[5][0], CASES[5][1], CASES[5][2], 20), (CASES[6][0], CASES[6][1], CASES[6][2], 7),
                                              ("launch_metric", lambda: problems.config("launch"), "perturb", 18),
                                              ("launch_ragged", _launch_ragged, "perturb", 150), ("bryson_denham", CASES[1][1], "perturb", 9)],
                         ids=[CASES[5][0], CASES[6][0], "launch_metric", "launch_ragged", "bryson_denham"])
def test_mfma_dx_mode_in_the_pipelined_kernel(built, name, make, mode, B):
    """dx_mode=1 in rpm_tile_pl_kernel (what bench.py runs): the DMA waves compute the tile's D.X with
    v_mfma_f64_16x16x4_f64 and publish it in LDS.  Defects agree with the scalar reference-order sum and with the oracle
    up to the summation order; every other output (path rows, events, linkages, the whole Jacobian) is bit-identical."""
    import torch
    prob = make()
    orc = oracle_for(prob)
    xl, xu, _, _ = orc.bounds()
    xs = np.stack([problems.seeded_iterate(orc.starting_point(), xl, xu, 70 + i, mode) for i in range(B)])
    dx = torch.from_numpy(xs).cuda()
    out = []
    for dxm in (0, 1):
        eng = NLPEngine(prob, n_instances=B, device=0, role_loop=1)
        eng.set_option("pipeline", 1)
        eng.set_option("dx_mode", dxm)
        dg = torch.full((B, eng.m), np.nan, dtype=torch.float64, device="cuda")
        dv = torch.full((B, eng.nnz_jac), np.nan, dtype=torch.float64, device="cuda")
        eng.eval_pair_dev(dx, dg, dv)
        dg2 = torch.full((B, eng.m), np.nan, dtype=torch.float64, device="cuda")
        eng.eval_g_dev(dx, dg2)
        torch.cuda.synchronize()
        assert eng.get_option("pipeline_active") == 1
        assert torch.equal(dg, dg2)
        out.append((dg.cpu().numpy(), dv.cpu().numpy()))
        eng.close()
    (g0, v0), (g1, v1) = out
    assert not np.isnan(g1).any()
    assert np.array_equal(v1, v0)
    # summation order only: |d_j x_j| reaches 1e3 on 1/64-wide intervals and 1e10 on the hp mesh's 1e-5-wide ones, where
    # defects of order 1 stand beside defects of order 1e8 — the bound is relative to the element or to the largest defect
    def close(a, b):
        return bool(np.all(np.abs(a - b) <= np.maximum(5e-12 * np.maximum(1.0, np.abs(b)), 1e-13 * np.abs(b).max())))
    assert close(g1, g0)
    for b in (0, B // 2, B - 1):
        assert close(g1[b], orc.eval_g(xs[b]))
    assert not np.array_equal(g1, g0) or name == "bryson_denham"   # the matrix cores really were used (different rounding)


@pytest.mark.gpu
@pytest.mark.parametrize("name,make,mode,B", [("launch_metric", lambda: problems.config("launch"), "perturb", 18), ("launch_ragged", _launch_ragged, "perturb", 150),
                                              ("quadrotor_8x8", lambda: problems.quadrotor(8, 8), "perturb", 300)],
                         ids=["launch_metric", "launch_ragged", "quadrotor_8x8"])
def test_staged_dynamics_in_the_pipelined_kernel(built, name, make, mode, B):
    """Option stage_roles (default -1: where the launch skips the constant block, persistent_values): functors that offer their dynamics in stages (problems.hpp has_stage: the launch vehicle,
    the quadrotor) are evaluated once per node in full and per perturbation role only in what the perturbed variable enters.
    Same operations on the same operands: g and every Jacobian value equal the whole-function evaluation's bit for bit, also
    with only g or only the Jacobian asked for."""
    import torch
    prob = make()
    orc = oracle_for(prob)
    xl, xu, _, _ = orc.bounds()
    xs = np.stack([problems.seeded_iterate(orc.starting_point(), xl, xu, 90 + i, mode) for i in range(B)])
    dx = torch.from_numpy(xs).cuda()
    out = []
    for staged in (1, 0):
        eng = NLPEngine(prob, n_instances=B, device=0, role_loop=1)
        eng.set_option("pipeline", 1)
        assert eng.get_option("stage_roles") == -1         # the default: staged only where the launch is bound by the dynamics
        eng.set_option("stage_roles", staged)
        dg = torch.full((B, eng.m), np.nan, dtype=torch.float64, device="cuda")
        dv = torch.full((B, eng.nnz_jac), np.nan, dtype=torch.float64, device="cuda")
        eng.eval_pair_dev(dx, dg, dv)
        dg2 = torch.full((B, eng.m), np.nan, dtype=torch.float64, device="cuda")
        dv2 = torch.full((B, eng.nnz_jac), np.nan, dtype=torch.float64, device="cuda")
        eng.eval_g_dev(dx, dg2)
        eng.eval_jac_g_dev(dx, dv2)
        torch.cuda.synchronize()
        assert eng.get_option("pipeline_active") == 1
        assert torch.equal(dg, dg2) and torch.equal(dv, dv2)
        out.append((dg.cpu().numpy(), dv.cpu().numpy()))
        eng.close()
    (g1, v1), (g0, v0) = out
    assert not np.isnan(g1).any() and not np.isnan(v1).any()
    assert np.array_equal(g1, g0) and np.array_equal(v1, v0)
    for b in (0, B - 1):
        assert np.max(np.abs(g1[b] - orc.eval_g(xs[b])) / np.maximum(1.0, np.abs(g0[b]))) <= 1e-12


# ---- exact Hessian (hessian-approximation=exact): forward second differences, LpHessian.cpp ------------------
def _exact():
    o = Options()
    o.SetStringValue("hessian-approximation", "exact")
    return o


HESS_CASES = [
    # (name, problem, relative tolerance).  Second differences divide rounding noise by h_i*h_j ~ 1e-12, so a 1-ulp
    # libm difference (exp/pow/sqrt/acos/sin/cos) between the CPU oracle and the GPU shows up at ~1e-4 of the
    # function's scale; polynomial dynamics have no libm call and must agree to rounding of the final combination.
    ("bryson_denham", lambda: problems.bryson_denham(3, 5), 1e-9),
    ("hypersensitive", lambda: problems.hypersensitive([-1, -0.5, 0.4, 1], [4, 6, 3], tf=50.0), 1e-9),
    ("brachistochrone", lambda: problems.brachistochrone(2, 6), 5e-3),
    ("quadrotor", lambda: problems.quadrotor(2, 4), 5e-3),
    ("climb", lambda: problems.min_time_climb(2, 6), 5e-3),
    ("launch", lambda: problems.launch(2, 5), 5e-3),
]


@pytest.mark.parametrize("name,make,tol", HESS_CASES, ids=[c[0] for c in HESS_CASES])
def test_exact_hessian(built, name, make, tol):
    import scipy.sparse as sp
    prob = make()
    eng = NLPEngine(prob, _exact(), device=0)
    orc = oracle_for(prob, _exact())
    assert eng.nnz_h == orc.nnz_h and eng.nnz_h > 0
    hi, hj = eng.eval_h_structure()
    oi, oj = orc.hess_structure()
    assert np.array_equal(hi, oi) and np.array_equal(hj, oj)
    assert np.all(hi >= hj)                                        # lower triangle
    xl, xu, _, _ = eng.get_bounds_info()
    x = problems.seeded_iterate(eng.get_starting_point(), xl, xu, 5)
    rng = np.random.RandomState(1)
    lam, sigma = rng.uniform(-1, 1, eng.m), 0.7
    hv, hr = eng.eval_h(x, sigma, lam), orc.eval_h(x, sigma, lam)
    scale = max(1.0, float(np.max(np.abs(hr))))
    assert np.max(np.abs(hv - hr)) <= tol * scale
    # the assembled matrix acts like the derivative of the Lagrangian gradient (oracle-free consistency check;
    # loose because of the reference's pertxf(i)-for-pertxf(j) denominators, LpHessian.cpp:1588,1612)
    if name in ("bryson_denham", "hypersensitive", "quadrotor"):
        H = sp.coo_matrix((hv, (hi, hj)), shape=(eng.n, eng.n)).tocsr()
        H = H + sp.tril(H, -1).T
        ji, jj = eng.eval_jac_g_structure()

        def grad_l(xx):
            J = sp.coo_matrix((eng.eval_jac_g(xx), (ji, jj)), shape=(eng.m, eng.n)).tocsr()
            return sigma * eng.eval_grad_f(xx) + J.T @ lam
        dx = rng.uniform(-1, 1, eng.n)
        e = 1e-5
        fd = (grad_l(x + e * dx) - grad_l(x - e * dx)) / (2 * e)
        assert np.max(np.abs(H @ dx - fd)) <= 2e-3 * max(1.0, np.max(np.abs(fd)))
    eng.close()


def test_exact_hessian_analytic_first_derivatives(built):
    opts = _exact()
    opts.SetStringValue("first-derive", "analytic")
    prob = problems.hypersensitive([-1, 0, 1], [5, 4], tf=30.0)
    eng, orc = NLPEngine(prob, opts, device=0), oracle_for(prob, opts)
    xl, xu, _, _ = eng.get_bounds_info()
    x = problems.seeded_iterate(eng.get_starting_point(), xl, xu, 2, "uniform")
    lam = np.linspace(-1, 1, eng.m)
    hv, hr = eng.eval_h(x, 1.3, lam), orc.eval_h(x, 1.3, lam)
    assert np.max(np.abs(hv - hr)) <= 1e-9 * max(1.0, np.max(np.abs(hr)))
    eng.close()


@pytest.mark.parametrize("name,make,mode", [CASES[3], CASES[5], CASES[6], CASES[7]], ids=[CASES[i][0] for i in (3, 5, 6, 7)])
def test_role_looped_layout_is_bit_identical(built, name, make, mode):
    """The throughput thread layout (64 nodes x 4 role groups, rpm_tile_rl_kernel) computes exactly the same numbers
    as the one-role-per-thread layout, for g, the Jacobian, batched instances and the analytic mode."""
    import torch
    prob = make()
    a = NLPEngine(prob, device=0, role_loop=0)
    b = NLPEngine(prob, device=0, role_loop=1)
    assert b.get_option("role_loop") == 1 and b.get_option("tile_nodes") == 64 and a.get_option("tile_nodes") == 16
    xl, xu, _, _ = a.get_bounds_info()
    x = problems.seeded_iterate(a.get_starting_point(), xl, xu, 23, mode)
    assert np.array_equal(a.eval_g(x), b.eval_g(x))
    assert np.array_equal(a.eval_jac_g(x, False), b.eval_jac_g(x, False))
    b.set_option("fuse_pair", 0)
    assert np.array_equal(a.eval_g(x), b.eval_g(x)) and np.array_equal(a.eval_jac_g(x), b.eval_jac_g(x))
    B = 5
    many = NLPEngine(prob, n_instances=B, device=0, role_loop=1)
    xs = np.stack([problems.seeded_iterate(a.get_starting_point(), xl, xu, 40 + i, mode) for i in range(B)])
    dg = torch.empty((B, a.m), dtype=torch.float64, device="cuda")
    dv = torch.empty((B, a.nnz_jac), dtype=torch.float64, device="cuda")
    many.eval_pair_dev(torch.from_numpy(xs).cuda(), dg, dv)
    torch.cuda.synchronize()
    for i in (0, B - 1):
        assert np.array_equal(dg[i].cpu().numpy(), a.eval_g(xs[i])) and np.array_equal(dv[i].cpu().numpy(), a.eval_jac_g(xs[i], False))
    for e_ in (a, b, many):
        e_.close()
    if name == "hypersensitive_hp":
        opts = Options()
        opts.SetStringValue("first-derive", "analytic")
        a2, b2 = NLPEngine(prob, opts, device=0, role_loop=0), NLPEngine(prob, opts, device=0, role_loop=1)
        assert np.array_equal(a2.eval_g(x), b2.eval_g(x)) and np.array_equal(a2.eval_jac_g(x), b2.eval_jac_g(x))
        a2.close()
        b2.close()


@pytest.mark.parametrize("name,make,mode,B", [(CASES[5][0], CASES[5][1], CASES[5][2], 20), (CASES[6][0], CASES[6][1], CASES[6][2], 7),
                                              (CASES[7][0], CASES[7][1], CASES[7][2], 3), ("launch_metric", lambda: problems.config("launch"), "perturb", 18),
                                              ("launch_ragged", _launch_ragged, "perturb", 150), ("climb_16x16", CASES[4][1], "perturb", 300),
                                              ("bryson_denham", CASES[1][1], "perturb", 9)],
                         ids=[CASES[5][0], CASES[6][0], CASES[7][0], "launch_metric", "launch_ragged", "climb_16x16", "bryson_denham"])
def test_pipelined_kernel_is_bit_identical(built, name, make, mode, B):
    """rpm_tile_pl_kernel (persistent workgroups: 4 compute waves + 1 DMA wave, endpoint items on the DMA waves) writes
    exactly what the role-looped kernel writes: g, Jacobian, every instance, whether a workgroup walks one tile or
    several (B is chosen so that some workgroups get one tile more than others), and in the g-only / J-only variants."""
    import torch
    prob = make()
    ref = NLPEngine(prob, n_instances=B, device=0, role_loop=1)
    pl = NLPEngine(prob, n_instances=B, device=0, role_loop=1)
    ref.set_option("pipeline", 0)
    pl.set_option("pipeline", 1)
    xl, xu, _, _ = ref.get_bounds_info()
    one = NLPEngine(prob, device=0)
    x0 = one.get_starting_point()
    one.close()
    xs = np.stack([problems.seeded_iterate(x0, xl, xu, 90 + i, mode) for i in range(B)])
    dx = torch.from_numpy(xs).cuda()
    out = []
    for eng in (ref, pl):
        dg = torch.full((B, ref.m), np.nan, dtype=torch.float64, device="cuda")
        dv = torch.full((B, ref.nnz_jac), np.nan, dtype=torch.float64, device="cuda")
        eng.eval_pair_dev(dx, dg, dv)
        dg2 = torch.full((B, ref.m), np.nan, dtype=torch.float64, device="cuda")
        dv2 = torch.full((B, ref.nnz_jac), np.nan, dtype=torch.float64, device="cuda")
        eng.eval_g_dev(dx, dg2)
        eng.eval_jac_g_dev(dx, dv2)
        torch.cuda.synchronize()
        assert eng.get_option("pipeline_active") == (1 if eng is pl else 0)
        out.append([t.cpu().numpy() for t in (dg, dv, dg2, dv2)])
    for a, b in zip(out[0], out[1]):
        assert not np.isnan(b).any()
        assert np.array_equal(a, b)
    assert np.array_equal(out[1][0], out[1][2]) and np.array_equal(out[1][1], out[1][3])
    ref.close()
    pl.close()


@pytest.mark.parametrize("layout", ["one_role", "role_looped", "pipelined"])
def test_persistent_values_skips_only_the_constant_block(built, layout):
    """Option persistent_values (SURVEY 8d's byte count B'): the second evaluation into a device `values` array this engine
    filled before writes everything that depends on x and leaves the constant Doffdiag block alone — the array equals a
    complete evaluation bit for bit; a marker the test plants inside that block survives (that the block is skipped is the
    contract, not an accident); another array, and the same array after any rpm_set_option, get everything."""
    import torch
    prob, B = _launch_ragged(), 9
    kw = {"one_role": dict(role_loop=0), "role_looped": dict(role_loop=1), "pipelined": dict(role_loop=1)}[layout]
    eng = NLPEngine(prob, n_instances=B, device=0, **kw)
    ref = NLPEngine(prob, n_instances=B, device=0, **kw)
    for e in (eng, ref):
        if layout != "one_role":
            e.set_option("pipeline", 1 if layout == "pipelined" else 0)
    eng.set_option("persistent_values", 1)
    one = NLPEngine(prob, device=0)
    xl, xu, _, _ = one.get_bounds_info()
    x0 = one.get_starting_point()
    nnz, nnz_const = one.nnz_jac, None
    i, j = one.eval_jac_g_structure()
    one.close()
    xs = [torch.from_numpy(np.stack([problems.seeded_iterate(x0, xl, xu, 200 + 10 * r + b) for b in range(B)])).cuda() for r in range(3)]
    dg = torch.empty((B, eng.m), dtype=torch.float64, device="cuda")
    dv = torch.full((B, nnz), np.nan, dtype=torch.float64, device="cuda")
    rv = torch.empty((B, nnz), dtype=torch.float64, device="cuda")
    eng.eval_pair_dev(xs[0], dg, dv)
    ref.eval_pair_dev(xs[0], dg, rv)
    torch.cuda.synchronize()
    assert torch.equal(dv, rv)
    const_entries = (dv[0] == dv[1]) & (dv[0] == dv[2])           # certainly contains the constant block
    marker_at = nnz - 5                                           # inside the Doffdiag copies (the CONST block is the tail)
    assert bool(const_entries[marker_at])
    dv[:, marker_at] = 4711.0
    torch.cuda.synchronize()
    eng.eval_pair_dev(xs[1], dg, dv)
    ref.eval_pair_dev(xs[1], dg, rv)
    torch.cuda.synchronize()
    assert bool((dv[:, marker_at] == 4711.0).all())
    dv[:, marker_at] = rv[:, marker_at]
    assert torch.equal(dv, rv)
    eng.eval_jac_g_dev(xs[2], dv)                                 # the Jacobian-only entry point takes the same shortcut
    ref.eval_jac_g_dev(xs[2], rv)
    other = torch.full((B, nnz), np.nan, dtype=torch.float64, device="cuda")
    eng.eval_jac_g_dev(xs[2], other)                              # an array the engine has not seen: everything
    torch.cuda.synchronize()
    assert torch.equal(dv, rv) and torch.equal(other, rv)
    dv[:, marker_at] = 4711.0
    eng.set_option("persistent_values", 1)                        # any rpm_set_option forgets the arrays
    torch.cuda.synchronize()
    eng.eval_pair_dev(xs[0], dg, dv)
    ref.eval_pair_dev(xs[0], dg, rv)
    torch.cuda.synchronize()
    assert torch.equal(dv, rv)
    eng.close()
    ref.close()


# ---- solution extraction (Nlp2OpConverter::Nlp2OpControl) --------------------------------
@pytest.mark.parametrize("name,make", [("launch", lambda: problems.launch(3, 6)), ("quadrotor", lambda: problems.quadrotor(4, 5)),
                                       ("hypersensitive", lambda: problems.config("hypersensitive")),
                                       ("bryson_denham", lambda: problems.bryson_denham())])
def test_solution_extraction(built, name, make, tmp_path):
    prob = make()
    eng, orc = NLPEngine(prob, device=0), oracle_for(prob)
    xl, xu, _, _ = eng.get_bounds_info()
    x = problems.seeded_iterate(eng.get_starting_point(), xl, xu, 31)
    lam = np.random.RandomState(3).uniform(-1, 1, eng.m)
    eng.finalize_solution(0, x, lam, 0.0)
    for ph in range(eng.n_phases):
        a, b = eng.nlp2op_control(ph), orc.nlp2op(ph, x, lam)      # a: from the stored solution
        for k in ("time", "state", "costate", "control", "pathmult"):
            assert np.array_equal(a[k], b[k]), (ph, k)               # copies, W^-1 lambda, spline extrapolation: bit-exact
        assert rel_err(a["hamiltonian"], b["hamiltonian"]) <= 1e-12  # dynamics (libm) inside
        assert a["mayer_cost"] == b["mayer_cost"]
        assert abs(a["lagrange_cost"] - b["lagrange_cost"]) <= 1e-13 * max(1.0, abs(b["lagrange_cost"]))
    eng.final_result_save(tmp_path)
    M = eng.phase_tables(0)["points"].size + 1
    st = np.loadtxt(tmp_path / "state1").reshape(M, -1)
    assert np.array_equal(st[:, 0], eng.nlp2op_control(0)["state"][:M])
    assert (tmp_path / ("Hamiltonian%d" % eng.n_phases)).exists() and (tmp_path / "costate1").exists()
    eng.close()


# ---- mesh-error estimate and ph refinement (SolutionErrorChecker / PhMeshRefineAlg) -----------
MESH_CASES = [("launch", lambda: problems.launch(3, 6)), ("launch_ragged", _launch_ragged),
              ("quadrotor", lambda: problems.quadrotor(4, 5)), ("hypersensitive", lambda: problems.config("hypersensitive")),
              ("bryson_denham", lambda: problems.bryson_denham()), ("brachistochrone", lambda: problems.config("brachistochrone")),
              ("climb", lambda: problems.config("climb"))]


@pytest.mark.parametrize("name,make", MESH_CASES, ids=[c[0] for c in MESH_CASES])
def test_solution_error_and_ph_refine(built, name, make):
    prob = make()
    eng, orc = NLPEngine(prob, device=0), oracle_for(prob)
    xl, xu, _, _ = eng.get_bounds_info()
    x = problems.seeded_iterate(eng.get_starting_point(), xl, xu, 77)
    eng.finalize_solution(0, x, np.zeros(eng.m), 0.0)
    for ph in range(eng.n_phases):
        a, b = eng.solution_error(ph), orc.solution_error(ph, x)     # a: from the stored solution
        assert a.shape == b.shape
        # interpolation, integration and the error quotient follow the oracle's operation order; the dynamics call
        # libm on both sides, so allow the same slack as the Hamiltonian of the extraction row
        assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(b).max()), (ph, np.abs(a - b).max())
        for tol, nmin, nmax in [(1e-6, 4, 16), (1e-3, 3, 8), (1e-9, 2, 5)]:
            d1, m1, n1, e1 = eng.ph_refine_mesh(ph, tol, nmin, nmax, x=x)
            d2, m2, n2, e2 = orc.ph_refine(ph, x, tol, nmin, nmax)
            assert d1 == d2 and np.array_equal(m1, m2) and np.array_equal(n1, n2), (ph, tol)
            assert np.abs(e1 - e2).max() <= 1e-12 * max(1.0, e2.max())
    eng.close()


def test_refinement_pass_builds_the_next_mesh(built):
    """One trip round the reference's outer loop without the NLP solve: estimate -> new mesh -> guess -> new engine."""
    from lpopc_amd.mesh import MeshRefiner, install_guess
    from lpopc_amd.problem import Options
    prob = problems.launch(3, 6)
    eng = NLPEngine(prob, device=0)
    xl, xu, _, _ = eng.get_bounds_info()
    x = problems.seeded_iterate(eng.get_starting_point(), xl, xu, 5)
    eng.finalize_solution(0, x, np.zeros(eng.m), 0.0)
    refiner = MeshRefiner(Options())
    assert refiner.RefineMesh(eng, prob) is False and refiner.CurrentGrid() == 1
    install_guess(eng, prob)
    old_n = eng.n
    eng.close()
    eng2, orc2 = NLPEngine(prob, device=0), oracle_for(prob)
    assert eng2.n != old_n and eng2.n == orc2.n
    x2 = eng2.get_starting_point()
    assert np.array_equal(x2, orc2.starting_point())               # the extracted solution, re-interpolated as the guess
    assert np.array_equal(eng2.eval_g(x2), orc2.eval_g(x2)) or rel_err(eng2.eval_g(x2), orc2.eval_g(x2)) <= 1e-12
    eng2.close()


def test_hpliu_refinement_end_to_end(built):
    """hp-Liu with the estimate computed on the device: same meshes as the oracle, call after call (the histories
    live in both objects), through the MeshRefiner mirror with mesh-refine-methods=hp-Liu."""
    from lpopc_amd.engine import HpLiuRefiner
    from lpopc_amd.mesh import MeshRefiner
    from lpopc_amd.problem import Options
    from oracle import oracle as orc_mod
    prob = problems.launch(3, 6)
    opts = Options()
    opts.SetStringValue("mesh-refine-methods", "hp-Liu")
    opts.SetNumericValue("desired-relative-error", 1e-4)
    refiner = MeshRefiner(opts)
    ho = orc_mod.HpLiu(prob.GetPhaseNum(), 1e-4, opts.GetIntegerValue("Nmax"), opts.GetNumericValue("R"))
    for it in range(3):
        eng, orc = NLPEngine(prob, device=0), oracle_for(prob)
        xl, xu, _, _ = eng.get_bounds_info()
        x = problems.seeded_iterate(eng.get_starting_point(), xl, xu, 11 + it)
        eng.finalize_solution(0, x, np.zeros(eng.m), 0.0)
        try:
            d_ref, m_ref = ho.refine(orc, x)
        except RuntimeError:
            with pytest.raises(Exception):
                refiner.RefineMesh(eng, prob)
            eng.close()
            break
        done = refiner.RefineMesh(eng, prob)
        eng.close()
        assert done == d_ref
        for i in range(prob.GetPhaseNum()):
            assert np.array_equal(prob.GetPhase(i).GetMeshPoints(), m_ref[i][0]), (it, i)
            assert list(prob.GetPhase(i).GetNodesPerInterval()) == [int(v) for v in m_ref[i][1]], (it, i)
        if done:
            break


def test_const_once_downloads_only_the_changing_prefix(built):
    """Option const_once: the second eval_jac_g into the same host array refreshes only the NL prefix; the array still
    equals a full evaluation, and a different array gets a full download again."""
    prob = problems.launch(3, 6)
    eng, full = NLPEngine(prob, device=0), NLPEngine(prob, device=0)
    eng.set_option("const_once", 1)
    xl, xu, _, _ = eng.get_bounds_info()
    x1 = problems.seeded_iterate(eng.get_starting_point(), xl, xu, 1)
    x2 = problems.seeded_iterate(eng.get_starting_point(), xl, xu, 2)
    buf = np.full(eng.nnz_jac, np.nan)
    eng.eval_jac_g(x1, out=buf)
    assert np.array_equal(buf, full.eval_jac_g(x1))
    nl = eng.nnz_jac - int(np.sum(buf == full.eval_jac_g(x2)))   # entries that change with x: all inside the NL prefix
    # the tail really stays at home: an entry of it the engine does not sample keeps what the caller scribbled there (the
    # option's contract is that the caller leaves the array alone), everything else equals a full evaluation
    k = eng.nnz_jac - 2
    tail_probe = buf[k]
    buf[k] = 12345.0
    eng.eval_jac_g(x2, out=buf)
    ref = full.eval_jac_g(x2)
    assert buf[k] == 12345.0 and np.array_equal(np.delete(buf, k), np.delete(ref, k)) and ref[k] == tail_probe and nl > 0
    # ... but an array whose sampled tail entries differ (the last entry is one: a fresh allocation at the same address
    # looks like this) is not trusted and gets everything
    buf[-1] = 777.0
    eng.eval_jac_g(x1, out=buf)
    assert np.array_equal(buf, full.eval_jac_g(x1))
    other = np.full(eng.nnz_jac, np.nan)
    eng.eval_jac_g(x2, out=other)
    assert np.array_equal(other, ref)
    eng.close()
    full.close()


def test_padded_instance_strides(built):
    """Option instance_align: in the device-resident batched calls every instance's g / values array starts on a
    multiple of that many doubles (rpm_get_option stride_g / stride_values); the numbers are those of the packed layout,
    the padding words are never written, with the pipelined, the role-looped and the one-role kernel."""
    import torch
    prob = problems.launch(3, 6)
    for role_loop, pipeline, B in ((0, 0, 5), (1, 0, 9), (1, 1, 11)):
        dense = NLPEngine(prob, n_instances=B, device=0, role_loop=role_loop)
        pad = NLPEngine(prob, n_instances=B, device=0, role_loop=role_loop)
        for e_ in (dense, pad):
            e_.set_option("pipeline", pipeline)
        pad.set_option("instance_align", 16)
        sg, sv = pad.get_option("stride_g"), pad.get_option("stride_values")
        assert sg % 16 == 0 and sv % 16 == 0 and sg >= pad.m and sv >= pad.nnz_jac and sg - pad.m < 16
        assert dense.get_option("stride_values") == dense.nnz_jac
        xl, xu, _, _ = dense.get_bounds_info()
        one = NLPEngine(prob, device=0)
        xs = np.stack([problems.seeded_iterate(one.get_starting_point(), xl, xu, 60 + i) for i in range(B)])
        one.close()
        dx = torch.from_numpy(xs).cuda()
        g0 = torch.empty((B, dense.m), dtype=torch.float64, device="cuda")
        v0 = torch.empty((B, dense.nnz_jac), dtype=torch.float64, device="cuda")
        g1 = torch.full((B, sg), 3.5, dtype=torch.float64, device="cuda")
        v1 = torch.full((B, sv), 3.5, dtype=torch.float64, device="cuda")
        dense.eval_pair_dev(dx, g0, v0)
        pad.eval_pair_dev(dx, g1, v1)
        torch.cuda.synchronize()
        assert pad.get_option("pipeline_active") == pipeline
        assert torch.equal(g1[:, :pad.m], g0) and torch.equal(v1[:, :pad.nnz_jac], v0)
        assert bool((g1[:, pad.m:] == 3.5).all()) and bool((v1[:, pad.nnz_jac:] == 3.5).all())
        g2 = torch.full((B, sg), 3.5, dtype=torch.float64, device="cuda")
        v2 = torch.full((B, sv), 3.5, dtype=torch.float64, device="cuda")
        pad.eval_g_dev(dx, g2)
        pad.eval_jac_g_dev(dx, v2)
        torch.cuda.synchronize()
        assert torch.equal(g2, g1) and torch.equal(v2, v1)
        dense.close()
        pad.close()



# ---- per-instance problem constants (parameter sweeps): rpm_set_instance_constants ------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["one_role", "role_looped", "pipelined"])
def test_instance_constants_bit_identical_to_separate_engines(built, layout):
    """A batched engine whose instances carry their own constants (here: the quadrotor's tracking target and weights)
    must produce, instance by instance, exactly what one-instance engines built with those constants produce — for every
    kernel layout and for f, grad f, g and the Jacobian (the batched Hessian is exercised by the sweep test in test_ipm.py)."""
    import torch
    B = 6
    rng = np.random.RandomState(11)
    prefs = [tuple(rng.uniform(-1.5, 1.5, size=3)) for _ in range(B)]
    probs = [problems.quadrotor(3, 5, pref=p) for p in prefs]
    eng = NLPEngine(probs[0], _exact(), n_instances=B, device=0, role_loop=0 if layout == "one_role" else 1)
    eng.set_option("pipeline", 1 if layout == "pipelined" else 0)
    for b in range(1, B):
        eng.set_instance_constants(b, probs[b].GetOpimalProblemFuns().consts)
    xl, xu, _, _ = eng.get_bounds_info()
    x0 = eng.get_starting_point()[:eng.n]
    xs = np.stack([problems.seeded_iterate(x0, xl, xu, 40 + b) for b in range(B)])
    d_x = torch.from_numpy(xs).cuda()
    d_g = torch.empty((B, eng.m), dtype=torch.float64, device="cuda")
    d_v = torch.empty((B, eng.nnz_jac), dtype=torch.float64, device="cuda")
    eng.eval_pair_dev(d_x, d_g, d_v)
    torch.cuda.synchronize()
    assert eng.get_option("pipeline_active") == (1 if layout == "pipelined" else 0)
    f = np.atleast_1d(eng.eval_f(xs.ravel()))
    grad = eng.eval_grad_f(xs.ravel()).reshape(B, eng.n)
    for b in range(B):
        one = NLPEngine(probs[b], _exact(), device=0)
        assert np.array_equal(d_g[b].cpu().numpy(), one.eval_g(xs[b]))
        assert np.array_equal(d_v[b].cpu().numpy(), one.eval_jac_g(xs[b], False))
        assert f[b] == one.eval_f(xs[b]) and np.array_equal(grad[b], one.eval_grad_f(xs[b]))
        one.close()
    assert len(set(np.round(f, 9))) == B                      # the constants really differ
    # wrong count / instance are refused
    with pytest.raises(Exception):
        eng.set_instance_constants(0, [1.0, 2.0])
    with pytest.raises(Exception):
        eng.set_instance_constants(B, probs[0].GetOpimalProblemFuns().consts)
    eng.close()

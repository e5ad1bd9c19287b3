"""The kernels bench.py times, at the batch sizes it times them, compared DIRECTLY with the CPU oracle (not with another
kernel layout), and the HIP path against the committed golden fixtures tests/golden/*.npz.

Tolerances as in tests/test_gpu_parity.py: eval_g / eval_f 1e-12 * max(1,|ref|); finite-difference eval_jac_g /
eval_grad_f 1e-8 * max(1,|ref|); constant and linear Jacobian entries, structure, bounds, starting point: bit-exact."""
import glob
import os

import numpy as np
import pytest

from lpopc_amd import problems
from lpopc_amd.engine import NLPEngine
from lpopc_amd.problem import Options

pytestmark = pytest.mark.gpu

G_TOL, JFD_TOL = 1e-12, 1e-8
HERE = os.path.dirname(os.path.abspath(__file__))


def rel_err(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))) if a.size else 0.0


BENCH = [
    # name, problem, iterates per launch (bench.py / tools/bench_configs.py), iterate mode
    ("config3_launch_B64", lambda: problems.launch(64, 16), 64, "perturb"),
    ("config4_hypersensitive_hp_B256", lambda: problems.config("hypersensitive"), 256, "uniform"),
    ("config5_quadrotor_B1024", lambda: problems.quadrotor(8, 8), 1024, "perturb"),
]


@pytest.mark.parametrize("name,make,B,mode", BENCH, ids=[c[0] for c in BENCH])
@pytest.mark.parametrize("kernel", ["pipelined", "role_looped"])
def test_bench_kernels_against_the_oracle(built, name, make, B, mode, kernel):
    import torch
    from oracle.oracle import Oracle
    prob = make()
    eng = NLPEngine(prob, n_instances=B, device=0, role_loop=1)
    eng.set_option("pipeline", 1 if kernel == "pipelined" else 0)
    eng.set_option("instance_align", 16)          # as bench.py: every instance's arrays start on a 128-byte line
    orc = Oracle(prob)
    xl, xu, _, _ = orc.bounds()
    x0 = orc.starting_point()
    xs = np.stack([problems.seeded_iterate(x0, xl, xu, 3 + i, mode) for i in range(B)])
    sg, sv = eng.get_option("stride_g"), eng.get_option("stride_values")
    d_x = torch.from_numpy(xs).cuda()
    d_g = torch.full((B, sg), np.nan, dtype=torch.float64, device="cuda")
    d_v = torch.full((B, sv), np.nan, dtype=torch.float64, device="cuda")
    eng.eval_pair_dev(d_x, d_g, d_v)
    d_f = torch.empty(B, dtype=torch.float64, device="cuda")
    d_grad = torch.zeros((B, eng.n), dtype=torch.float64, device="cuda")
    eng.eval_f_dev(d_x, d_f)
    eng.eval_grad_f_dev(d_x, d_grad)
    torch.cuda.synchronize()
    assert eng.get_option("pipeline_active") == (1 if kernel == "pipelined" else 0)
    assert eng.get_option("role_loop") == 1
    # every slot of every instance written (checked on the device: the batch is up to 436 MB)
    assert not bool(torch.isnan(d_g[:, :eng.m]).any()) and not bool(torch.isnan(d_v[:, :eng.nnz_jac]).any())
    n_const = sum(orc.phase_tables(p)["doff_vals"].size * prob.GetPhase(p).get_optimal_info()[0] for p in range(eng.n_phases))
    f = d_f.cpu().numpy()
    picks = sorted(set([0, 1, B // 7, B // 3, B // 2, B - B // 5, B - 2, B - 1]))
    assert len(picks) >= 8
    for b in picks:
        g_b, v_b, grad_b = d_g[b, :eng.m].cpu().numpy(), d_v[b, :eng.nnz_jac].cpu().numpy(), d_grad[b].cpu().numpy()
        g_ref, v_ref = orc.eval_g(xs[b]), orc.eval_jac_g(xs[b])
        assert rel_err(g_b, g_ref) <= G_TOL, (b, rel_err(g_b, g_ref))
        assert rel_err(v_b, v_ref) <= JFD_TOL, (b, rel_err(v_b, v_ref))
        assert np.array_equal(v_b[eng.nnz_jac - n_const:], v_ref[-n_const:])      # constant block: copies
        f_ref = orc.eval_f(xs[b])
        assert abs(f[b] - f_ref) <= 1e-12 * max(1.0, abs(f_ref))
        assert rel_err(grad_b, orc.eval_grad_f(xs[b])) <= JFD_TOL
    eng.close()


def test_batched_hessian_against_the_oracle(built):
    """eval_h of a batch (what the device interior-point solver consumes), instance by instance against the oracle."""
    import torch
    from oracle.oracle import Oracle
    opts = Options()
    opts.SetStringValue("hessian-approximation", "exact")
    B = 9
    prob = problems.quadrotor(8, 8)
    eng = NLPEngine(prob, opts, n_instances=B, device=0)
    orc = Oracle(prob, opts)
    xl, xu, _, _ = orc.bounds()
    xs = np.stack([problems.seeded_iterate(orc.starting_point(), xl, xu, 50 + i) for i in range(B)])
    lam = np.random.RandomState(4).uniform(-1, 1, (B, eng.m))
    d_h = torch.full((B, eng.nnz_h), np.nan, dtype=torch.float64, device="cuda")
    eng.eval_h_dev(torch.from_numpy(xs).cuda(), 0.7, torch.from_numpy(lam).cuda(), d_h)
    torch.cuda.synchronize()
    h = d_h.cpu().numpy()
    for b in range(B):
        hr = orc.eval_h(xs[b], 0.7, lam[b])
        # second differences: a 1-ulp sin/cos difference divided by h_a h_b ~ 1e-12 (tests/test_gpu_parity.py HESS_CASES)
        assert np.max(np.abs(h[b] - hr)) <= 5e-3 * max(1.0, float(np.max(np.abs(hr)))), b
    eng.close()


GOLDEN = {
    "brachistochrone_1x10": lambda: problems.brachistochrone(1, 10),
    "bryson_denham_1x20": lambda: problems.bryson_denham(),
    "launch_1x20": lambda: problems.launch(),
    "launch_4x8": lambda: problems.launch(4, 8),
    "hypersensitive_6x5": lambda: problems.hypersensitive([-1, -0.8, -0.3, 0.2, 0.7, 0.9, 1], [5] * 6),
    "climb_4x6": lambda: problems.min_time_climb(4, 6),
    "quadrotor_2x5": lambda: problems.quadrotor(2, 5),
}


def test_every_committed_fixture_has_a_case():
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(HERE, "golden", "*.npz")))
    assert names == sorted(GOLDEN)


@pytest.mark.parametrize("name", sorted(GOLDEN))
@pytest.mark.parametrize("layout", ["one_role", "role_looped", "pipelined"])
def test_hip_path_reproduces_the_golden_fixtures(built, name, layout):
    """The committed vectors (tests/golden/make_golden.py; produced by the oracle, see its provenance note) through the
    C ABI on the GPU: structure, bounds and starting point bit-exact, values within the stated tolerances."""
    fx = np.load(os.path.join(HERE, "golden", name + ".npz"))
    eng = NLPEngine(GOLDEN[name](), device=0, role_loop=0 if layout == "one_role" else 1)
    eng.set_option("pipeline", 1 if layout == "pipelined" else 0)
    assert (eng.n, eng.m, eng.nnz_jac) == (fx["x"].size, fx["g"].size, fx["jac_values"].size)
    i, j = eng.eval_jac_g_structure()
    assert np.array_equal(i, fx["jac_i"]) and np.array_equal(j, fx["jac_j"])
    xl, xu, gl, gu = eng.get_bounds_info()
    for a, b in ((xl, "x_l"), (xu, "x_u"), (gl, "g_l"), (gu, "g_u"), (eng.get_starting_point(), "x_start")):
        assert np.array_equal(a, fx[b]), b
    x = fx["x"]
    assert rel_err(eng.eval_g(x), fx["g"]) <= G_TOL
    assert rel_err(eng.eval_jac_g(x, False), fx["jac_values"]) <= JFD_TOL
    assert abs(eng.eval_f(x) - fx["f"][0]) <= 1e-12 * max(1.0, abs(fx["f"][0]))
    assert rel_err(eng.eval_grad_f(x), fx["grad_f"]) <= JFD_TOL
    eng.close()

"""rpm_group_* (include/rpm_hip.h): ONE process, several GPUs — the mesh intervals of one NLP sharded over the listed
devices behind the C ABI, for lpopc's single-process caller (Core/LpNLPSolver.cpp:13-53).  On the one-GPU box the same
device is listed 8 times: the sharding, the shared page-locked arrays, the delivery by difference and the peer push are
exercised exactly as on 8 devices (only the links are not).  Bar: bit-identical to a single engine's vectors."""
import numpy as np
import pytest

from lpopc_amd import problems
from lpopc_amd.engine import NLPEngine, RpmError
from lpopc_amd.group import EngineGroup


def _own_pages(n):
    import mmap
    return np.frombuffer(mmap.mmap(-1, max(8 * n, 8)), dtype=np.float64, count=n)


def _iterates(eng, count, mode):
    xl, xu, _, _ = eng.get_bounds_info()
    x0 = eng.get_starting_point()
    return [problems.seeded_iterate(x0, xl, xu, 60 + i, mode) for i in range(count)]


def test_group_set_up_is_host_only_and_fails_loudly_without_a_device(built):
    import torch
    prob = problems.config("hypersensitive")
    one = NLPEngine(prob)
    grp = EngineGroup(prob, [0] * 8)
    assert grp.size == 8 and (grp.n, grp.m, grp.nnz_jac) == (one.n, one.m, one.nnz_jac)
    i, j = grp.eval_jac_g_structure()
    i1, j1 = one.eval_jac_g_structure()
    assert np.array_equal(i, i1) and np.array_equal(j, j1)
    # the ranks' shares partition the vectors (rank 0 also owns the endpoint rows)
    for which, size in ((0, one.m), (1, one.nnz_jac)):
        seen = np.zeros(size, dtype=np.int32)
        for r in range(8):
            e = NLPEngine(prob, shard_mode=1, shard_rank=r, shard_world=8)
            for off, ln, _ in e.shard_segments(which, r)[0]:
                seen[off:off + ln] += 1
            e.close()
        assert (seen == 1).all()
    assert grp.engine(3).get_option("pin_host") == 1 and grp.engine(3).get_option("delta_values") == 1
    if not torch.cuda.is_available():
        x, g = _own_pages(grp.n), _own_pages(grp.m)
        with pytest.raises(RpmError) as ei:
            grp.eval_g(x, g)
        assert "no CPU fallback" in str(ei.value) and "rank 0" in str(ei.value)
    with pytest.raises(RpmError):
        EngineGroup(prob, [0] * 17)
    grp.close()
    one.close()


CASES = [("hypersensitive_hp", lambda: problems.config("hypersensitive"), "uniform", 8),
         ("launch_4x16x8", lambda: problems.launch(16, 8), "perturb", 8),
         ("launch_metric", lambda: problems.launch(64, 16), "perturb", 3)]


@pytest.mark.gpu
@pytest.mark.parametrize("name,make,mode,world", CASES, ids=[c[0] for c in CASES])
def test_host_consumer_group_equals_a_single_engine(built, name, make, mode, world):
    prob = make()
    one = NLPEngine(prob, device=0)
    grp = EngineGroup(prob, [0] * world)
    xs = _iterates(one, 4, mode)
    x, g, v, gr = _own_pages(one.n), _own_pages(one.m), _own_pages(one.nnz_jac), _own_pages(one.n)
    g[:] = np.nan
    v[:] = np.nan
    for k in (0, 1, 2, 1, 1, 3):
        x[:] = xs[k]
        ref_g, ref_v = one.eval_pair(xs[k])
        grp.eval_g(x, g, True)                      # the two TNLP calls Ipopt makes per iterate
        grp.eval_jac_g(x, v, False)
        assert np.array_equal(g, ref_g) and np.array_equal(v, ref_v), k
        assert grp.eval_f(x, False) == one.eval_f(xs[k]) and np.array_equal(grp.eval_grad_f(x, gr, False), one.eval_grad_f(xs[k]))
        g[:] = 0.0
        grp.eval_pair(x, g, v)                      # ... or one call for both
        assert np.array_equal(g, ref_g) and np.array_equal(v, ref_v), k
    # all ranks share ONE registration per array; by difference: later deliveries were partial
    e0 = grp.engine(0)
    assert e0.get_option("pin_live") == 4 and e0.get_option("pin_held") == 4
    sent = sum(grp.engine(r).get_option("delta_sent_runs") for r in range(world))
    total = sum(grp.engine(r).get_option("delta_total_runs") for r in range(world))
    assert 0 < total <= sent < 12 * total
    # a NaN among the nodes of the LAST rank: the call fails, says which rank, and the next call is complete again
    x[:] = xs[0]
    x[one.n - 3] = np.nan
    with pytest.raises(RpmError) as ei:
        grp.eval_pair(x, g, v)
    assert "non-finite" in str(ei.value) and "rank %d" % (world - 1) in str(ei.value)
    x[:] = xs[0]
    grp.eval_pair(x, g, v)
    ref_g, ref_v = one.eval_pair(xs[0])
    assert np.array_equal(g, ref_g) and np.array_equal(v, ref_v)
    grp.close()
    assert one.get_option("pin_live") == 0
    one.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,make,mode,world", CASES[:2], ids=[c[0] for c in CASES[:2]])
def test_device_consumer_group_direct_stores_and_peer_push(built, name, make, mode, world):
    import torch
    prob = make()
    B = 2
    one = NLPEngine(prob, n_instances=B, device=0)
    grp = EngineGroup(prob, [0] * world, n_instances=B)
    xs = np.stack(_iterates(NLPEngine(prob), B, mode))
    d_x = torch.from_numpy(xs).cuda().reshape(-1)
    sg, sv = one.get_option("stride_g"), one.get_option("stride_values")
    ref_g = torch.zeros(B * sg, dtype=torch.float64, device="cuda")
    ref_v = torch.zeros(B * sv, dtype=torch.float64, device="cuda")
    one.eval_pair_dev(d_x, ref_g, ref_v)
    torch.cuda.synchronize()
    # every rank's tile kernel stores straight into the home rank's arrays
    d_g = torch.full((B * sg,), float("nan"), dtype=torch.float64, device="cuda")
    d_v = torch.full((B * sv,), float("nan"), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    grp.eval_pair_dev(0, d_x, d_g, d_v)
    for b in range(B):
        assert torch.equal(d_g[b * sg:b * sg + one.m], ref_g[b * sg:b * sg + one.m])
        assert torch.equal(d_v[b * sv:b * sv + one.nnz_jac], ref_v[b * sv:b * sv + one.nnz_jac])
    # all-gather: every rank's own arrays complete after ONE push per rank
    gs = [torch.full((B * sg,), float("nan"), dtype=torch.float64, device="cuda") for _ in range(world)]
    vs = [torch.full((B * sv,), float("nan"), dtype=torch.float64, device="cuda") for _ in range(world)]
    torch.cuda.synchronize()
    grp.allgather_pair_dev([d_x] * world, gs, vs)
    for r in range(world):
        for b in range(B):
            assert torch.equal(gs[r][b * sg:b * sg + one.m], ref_g[b * sg:b * sg + one.m]), r
            assert torch.equal(vs[r][b * sv:b * sv + one.nnz_jac], ref_v[b * sv:b * sv + one.nnz_jac]), r
    grp.close()
    one.close()


@pytest.mark.gpu
def test_sweep_group_equals_one_engine(built):
    """rpm_sweep_*: the instances of an MPC sweep dealt to three shares (this box's one GPU listed three times: three engines,
    three solvers, three host threads) — every instance's solution, multipliers, verdict and iteration count are bit for bit
    what ONE engine holding all of them computes."""
    from lpopc_amd.engine import BatchedIPM, NLPEngine
    from lpopc_amd.group import SweepGroup
    from lpopc_amd.problem import Options
    o = Options()
    o.SetStringValue("hessian-approximation", "exact")
    B = 7
    prob = problems.quadrotor(4, 6)
    one = NLPEngine(prob, o, n_instances=B, device=0)
    xl, xu, _, _ = one.get_bounds_info()
    x0 = np.tile(one.get_starting_point()[:one.n], (B, 1))
    N1 = 4 * 6 + 1
    idx = [i * N1 for i in range(12)]
    rng = np.random.RandomState(2)
    bounds = []
    for bi in range(B):
        l, u = xl.copy(), xu.copy()
        l[idx] = u[idx] = rng.uniform(-0.3, 0.3, 12)
        bounds.append((l, u))
    ipm = BatchedIPM(one, max_iter=200)
    for bi in range(B):
        ipm.set_bounds(bi, *bounds[bi])
    ref = ipm.solve(x0)
    ipm.close()
    one.close()
    sw = SweepGroup(prob, [0, 0, 0], B, o, max_iter=200)
    assert sw.size == 3 and sw.shares() == [(0, 2), (2, 2), (4, 3)]
    for bi in range(B):
        sw.set_bounds(bi, *bounds[bi])
    r = sw.solve(x0)
    assert (ref["status"] == 0).all() and np.array_equal(r["status"], ref["status"])
    assert np.array_equal(r["iterations"], ref["iterations"])
    assert np.array_equal(r["x"], ref["x"]) and np.array_equal(r["lambda"], ref["lambda"]) and np.array_equal(r["obj"], ref["obj"])
    assert sw.stats()["iterations"] == int(ref["iterations"].max())
    with pytest.raises(Exception):
        sw.set_bounds(B, *bounds[0])                     # no such instance
    sw.close()
    with pytest.raises(Exception):
        SweepGroup(prob, [0, 0, 0], 2, o)               # fewer instances than devices

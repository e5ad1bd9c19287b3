"""Host-pointer (Ipopt-facing) path on the GPU: every delivery variant of rpm_eval_g / rpm_eval_jac_g / rpm_eval_pair
gives bit-identical arrays — copy-engine staging (the round-1 path), page-locked zero-copy x / g, NL-prefix-only
("const_once"), delivery by difference ("delta_values"), one call or two — and equals the CPU oracle within the
tolerances of tests/test_gpu_parity.py.  Reference behaviour: Core/LpopcIpopt.cpp:135-181 (copy x in, call
GetAllCons / GetConsJacbi, copy the result out)."""
import numpy as np
import pytest

from lpopc_amd import problems
from lpopc_amd.engine import NLPEngine

pytestmark = pytest.mark.gpu

G_TOL, JFD_TOL = 1e-12, 1e-8


def rel_err(a, b):
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))) if a.size else 0.0


def _iterates(eng, count, mode="perturb"):
    xl, xu, _, _ = eng.get_bounds_info()
    x0 = eng.get_starting_point()
    return [problems.seeded_iterate(x0, xl, xu, 100 + i, mode) for i in range(count)]


@pytest.fixture(autouse=True)
def _registrations_are_released(built):
    """Every page-lock registration an engine makes is released by rpm_destroy / eviction: a refused hipHostUnregister
    would leave a stale pinned range behind for whatever the process allocates at that address next."""
    yield
    probe = NLPEngine(problems.brachistochrone(1, 10))
    failures, made = probe.get_option("pin_unregister_failures"), probe.get_option("pin_registered")
    probe.close()
    assert failures == 0, (failures, made)


CASES = [
    ("launch_3x6", lambda: problems.launch(3, 6), "perturb"),                 # arrays below the 64 KB pinning threshold
    ("launch_metric", lambda: problems.launch(64, 16), "perturb"),
    ("quadrotor_8x8", lambda: problems.quadrotor(8, 8), "perturb"),
    ("hypersensitive_hp", lambda: problems.config("hypersensitive"), "uniform"),
]


@pytest.mark.parametrize("name,make,mode", CASES, ids=[c[0] for c in CASES])
def test_delivery_variants_are_bit_identical(built, name, make, mode):
    from oracle.oracle import Oracle
    prob = make()
    staged = NLPEngine(prob, device=0)           # wrapper default: pin_host = 0 -> copy-engine staging of x, g, values
    staged.set_option("zero_copy", 0)
    xs = _iterates(staged, 4, mode)
    ref = [(staged.eval_g(x, True).copy(), staged.eval_jac_g(x, False).copy()) for x in xs]
    orc = Oracle(prob)
    assert rel_err(ref[0][0], orc.eval_g(xs[0])) <= G_TOL and rel_err(ref[0][1], orc.eval_jac_g(xs[0])) <= JFD_TOL

    def fresh(**opts):
        e = NLPEngine(prob, device=0)
        e.set_option("pin_host", 1)
        for k, v in opts.items():
            e.set_option(k, v)
        return e

    # caller-owned arrays handed again and again, as Ipopt's TNLPAdapter does
    for opts in ({}, {"const_once": 1}, {"delta_values": 1}, {"delta_values": 1, "zero_copy": 0}):
        e = fresh(**opts)
        xbuf, g, v = np.zeros(e.n), np.full(e.m, np.nan), np.full(e.nnz_jac, np.nan)
        for i, x in enumerate(xs):
            xbuf[:] = x
            e.eval_g(xbuf, True, out=g)
            e.eval_jac_g(xbuf, False, out=v)
            assert np.array_equal(g, ref[i][0]), (opts, i)
            assert np.array_equal(v, ref[i][1]), (opts, i)
        # eval_jac_g without a preceding eval_g of the same x (new_x = true)
        xbuf[:] = xs[1]
        e.eval_jac_g(xbuf, True, out=v)
        assert np.array_equal(v, ref[1][1]), opts
        # rpm_eval_pair into the same arrays
        for i in (2, 0):
            xbuf[:] = xs[i]
            e.eval_pair(xbuf, g, v)
            assert np.array_equal(g, ref[i][0]) and np.array_equal(v, ref[i][1]), (opts, i)
        # a different values array gets a complete delivery
        v2 = np.full(e.nnz_jac, np.nan)
        e.eval_jac_g(xbuf, False, out=v2)
        assert np.array_equal(v2, ref[0][1]), opts
        e.close()
    staged.close()


def test_delta_delivery_sends_only_what_changed_and_notices_a_disturbed_array(built):
    prob = problems.launch(64, 16)
    full = NLPEngine(prob, device=0)
    e = NLPEngine(prob, device=0)
    e.set_option("pin_host", 1)
    e.set_option("delta_values", 1)
    xs = _iterates(e, 3)
    xbuf, g, v = np.zeros(e.n), np.zeros(e.m), np.full(e.nnz_jac, np.nan)
    xbuf[:] = xs[0]
    e.eval_pair(xbuf, g, v)
    total = e.get_option("delta_total_runs")
    assert e.get_option("delta_sent_runs") == total and total == -(-e.nnz_jac // 512)    # first delivery: everything
    xbuf[:] = xs[1]
    e.eval_pair(xbuf, g, v)
    sent = e.get_option("delta_sent_runs")
    # the constant Doffdiag block (54 % of the entries), the linear entries and the x-independent finite-difference
    # blocks of the launch dynamics stay at home
    assert 0 < sent < 0.5 * total, (sent, total)
    assert np.array_equal(v, full.eval_jac_g(xs[1]))
    # same x again: nothing changed, nothing is sent
    e.eval_pair(xbuf, g, v)
    assert e.get_option("delta_sent_runs") == 0 and np.array_equal(v, full.eval_jac_g(xs[1]))
    # the caller disturbs the array where the engine samples it (first entry): complete re-delivery
    v[0] = 4711.0
    v[12345] = -1.0
    xbuf[:] = xs[2]
    e.eval_pair(xbuf, g, v)
    assert e.get_option("delta_sent_runs") == total
    assert np.array_equal(v, full.eval_jac_g(xs[2])) and np.array_equal(g, full.eval_g(xs[2]))
    e.close()
    full.close()


def test_batched_host_path(built):
    """n_instances > 1 through the host-pointer entry points (instance-major arrays), all delivery variants."""
    B = 5
    prob = problems.quadrotor(8, 8)
    one = NLPEngine(prob, device=0)
    xs = np.stack(_iterates(one, B))
    ref_g = np.concatenate([one.eval_g(x) for x in xs])
    ref_v = np.concatenate([one.eval_jac_g(x) for x in xs])
    for opts in ({}, {"delta_values": 1}):
        e = NLPEngine(prob, n_instances=B, device=0)
        e.set_option("pin_host", 1)
        for k, val in opts.items():
            e.set_option(k, val)
        xbuf, g, v = xs.ravel().copy(), np.zeros(B * e.m), np.zeros(B * e.nnz_jac)
        e.eval_pair(xbuf, g, v)
        assert np.array_equal(g, ref_g) and np.array_equal(v, ref_v)
        e.eval_g(xbuf, True, out=g)
        e.eval_jac_g(xbuf, False, out=v)
        assert np.array_equal(g, ref_g) and np.array_equal(v, ref_v)
        e.close()
    one.close()


def test_nonfinite_is_reported_by_every_entry_point(built):
    prob = problems.launch(64, 16)
    for pin in (0, 1):
        e = NLPEngine(prob, device=0)
        e.set_option("pin_host", pin)
        x = e.get_starting_point()
        x[5] = np.inf
        g, v = np.zeros(e.m), np.zeros(e.nnz_jac)
        for call in (lambda: e.eval_g(x, True, out=g), lambda: e.eval_jac_g(x, True, out=v), lambda: e.eval_pair(x, g, v)):
            with pytest.raises(Exception) as ei:
                call()
            assert "non-finite" in str(ei.value)
        e.set_option("check_finite", 0)           # lpopc's behaviour: the NaNs are handed to the caller
        e.eval_pair(x, g, v)
        assert not np.isfinite(g).all()
        e.close()


def test_many_caller_arrays_do_not_accumulate_registrations(built):
    """pin_host keeps at most 8 page-locked registrations (LRU): 20 distinct arrays in a row still evaluate correctly."""
    prob = problems.launch(64, 16)
    e = NLPEngine(prob, device=0)
    e.set_option("pin_host", 1)
    x = _iterates(e, 1)[0]
    ref_g, ref_v = e.eval_g(x).copy(), e.eval_jac_g(x).copy()
    keep = []
    for i in range(20):
        g, v = np.zeros(e.m), np.zeros(e.nnz_jac)
        keep.append((g, v))
        e.eval_pair(x, g, v)
        assert np.array_equal(g, ref_g) and np.array_equal(v, ref_v)
    e.close()


def test_objective_and_gradient_cache_follows_new_x(built):
    """eval_f / eval_grad_f share one launch per x (LpopcIpopt::eval_f / eval_grad_f, Core/LpopcIpopt.cpp:106-133); any
    callback that receives new_x = true drops what the others cached, whichever callback Ipopt happens to call first."""
    for prob in (problems.quadrotor(8, 8), problems.launch(64, 16)):
        ref = NLPEngine(prob, device=0)
        e = NLPEngine(prob, device=0)
        e.set_option("pin_host", 1)
        x1, x2, x3 = _iterates(e, 3)
        f = [ref.eval_f(x) for x in (x1, x2, x3)]
        gr = [ref.eval_grad_f(x).copy() for x in (x1, x2, x3)]
        assert len({float(v) for v in f}) == 3
        # pin_host = 1: caller-owned arrays that outlive the calls (the registration contract of rpm_hip.h)
        gb, cb, vb = np.zeros(e.n), np.zeros(e.m), np.zeros(e.nnz_jac)
        assert e.eval_f(x1, True) == f[0] and np.array_equal(e.eval_grad_f(x1, False, out=gb), gr[0])
        e.eval_g(x2, True, out=cb)                                # a constraint callback sees the new x first
        assert np.array_equal(cb, ref.eval_g(x2))
        assert e.eval_f(x2, False) == f[1] and np.array_equal(e.eval_grad_f(x2, False, out=gb), gr[1])
        assert np.array_equal(e.eval_grad_f(x3, True, out=gb), gr[2]) and e.eval_f(x3, False) == f[2]
        assert np.array_equal(e.eval_jac_g(x3, False, out=vb), ref.eval_jac_g(x3))   # no cached pair for x3: evaluated from x
        xbad = x1.copy()
        xbad[e.n - 1] = np.nan                                    # tf of the last phase
        with pytest.raises(Exception) as ei:
            e.eval_grad_f(xbad, True, out=gb)
        assert "non-finite" in str(ei.value)
        e.close()
        ref.close()

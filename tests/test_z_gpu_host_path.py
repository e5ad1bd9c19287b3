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


def _own_pages(n):
    """A float64 array on pages of its own (an anonymous mapping): what malloc hands out may share its first and last page
    with a neighbour, which changes how many regions the registry needs."""
    import mmap
    return np.frombuffer(mmap.mmap(-1, max(8 * n, 8)), dtype=np.float64, count=n)


def _iterates(eng, count, mode="perturb"):
    xl, xu, _, _ = eng.get_bounds_info()
    x0 = eng.get_starting_point()
    return [problems.seeded_iterate(x0, xl, xu, 100 + i, mode) for i in range(count)]


@pytest.fixture(autouse=True)
def _registrations_are_released(built):
    """Every page-lock registration is gone once its last holder is: after each test the process-wide table (librpm_pin.so)
    is empty and no hipHostRegister / hipHostUnregister was refused behind the caller's back."""
    probe = NLPEngine(problems.brachistochrone(1, 10))
    before = {k: probe.get_option(k) for k in ("pin_unregister_failures", "pin_register_failures")}
    yield
    after = {k: probe.get_option(k) for k in before}
    live, made, gone = (probe.get_option(k) for k in ("pin_live", "pin_registered", "pin_unregistered"))
    probe.close()
    assert after == before, (before, after, probe.last_error() if probe._h else "")
    assert live == 0 and made == gone, (live, made, gone)


CASES = [
    ("launch_3x6", lambda: problems.launch(3, 6), "perturb"),                 # g below the 64 KB registration threshold: staged
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
    """An engine holds at most 8 page-locked registrations (least recently used is let go first): 20 distinct arrays in a
    row still evaluate correctly, and what it let go of was unregistered."""
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
        assert e.get_option("pin_held") <= 8
        assert e.get_option("pin_live_kb") <= 8 * (8 * e.nnz_jac // 1024 + 8)      # never more than 8 arrays' pages
    assert e.get_option("pin_evicted") > 0 and e.get_option("pin_live") <= 8
    e.close()


def test_pin_host_is_opt_in_and_the_default_path_never_registers_caller_memory(built):
    """C ABI default: "pin_host" = 0 — x / g / values / grad_f go through the engine's own page-locked staging buffers, the
    caller's arrays are never registered (they may have any lifetime), results are bit-identical to the registered path."""
    prob = problems.launch(64, 16)
    e, p = NLPEngine(prob, device=0), NLPEngine(prob, device=0)
    assert e.get_option("pin_host") == 0
    p.set_option("pin_host", 1)
    made = e.get_option("pin_registered")
    xs = _iterates(e, 3)
    xbuf, g, v, gr = _own_pages(e.n), _own_pages(e.m), _own_pages(e.nnz_jac), _own_pages(e.n)
    for zc in (1, 0):
        e.set_option("zero_copy", zc)
        for x in xs:
            before = e.get_option("pin_registered")
            got = (e.eval_g(x + 0.0, True), e.eval_jac_g(x + 0.0, False), e.eval_f(x + 0.0, True), e.eval_grad_f(x + 0.0, False))   # temporaries
            assert e.get_option("pin_registered") == before and e.get_option("pin_held") == 0
            xbuf[:] = x
            p.eval_g(xbuf, True, out=g)
            p.eval_jac_g(xbuf, False, out=v)
            f = p.eval_f(xbuf, True)
            p.eval_grad_f(xbuf, False, out=gr)
            assert np.array_equal(got[0], g) and np.array_equal(got[1], v) and got[2] == f and np.array_equal(got[3], gr)
            g2, v2 = e.eval_pair(x + 0.0)
            assert np.array_equal(g2, g) and np.array_equal(v2, v)
    assert p.get_option("pin_held") == 4 and p.get_option("pin_registered") == made + 4
    e.close()
    p.close()


def test_page_lock_registry_is_process_wide_page_granular_and_reference_counted(built):
    prob = problems.launch(64, 16)
    ref = NLPEngine(prob, device=0)
    x = _iterates(ref, 1)[0]
    ref_g, ref_v = ref.eval_pair(x)
    a, b = NLPEngine(prob, device=0), NLPEngine(prob, device=0)
    for e in (a, b):
        e.set_option("pin_host", 1)
    xbuf, g, v = _own_pages(a.n), _own_pages(a.m), _own_pages(a.nnz_jac)
    xbuf[:] = x
    made, shared = a.get_option("pin_registered"), a.get_option("pin_shared")
    a.eval_pair(xbuf, g, v)
    assert a.get_option("pin_registered") == made + 3 and a.get_option("pin_live") == 3
    # a second engine asks for the same arrays: no second hipHostRegister, it shares the three registrations
    g[:] = 0
    v[:] = 0
    b.eval_pair(xbuf, g, v)
    assert np.array_equal(g, ref_g) and np.array_equal(v, ref_v)
    assert b.get_option("pin_registered") == made + 3 and b.get_option("pin_shared") == shared + 3 and b.get_option("pin_live") == 3
    # rpm_destroy of one holder does not unpin pages the other still addresses
    a.close()
    assert b.get_option("pin_live") == 3 and b.get_option("pin_held") == 3
    g[:] = 0
    v[:] = 0
    b.eval_pair(xbuf, g, v)
    assert np.array_equal(g, ref_g) and np.array_equal(v, ref_v)
    # "pin_host" = 0 lets go of everything at once (what a caller does before freeing its arrays)
    b.set_option("pin_host", 0)
    assert b.get_option("pin_live") == 0 and b.get_option("pin_held") == 0
    b.set_option("pin_host", 1)
    # two arrays that share a page (one allocation, the boundary in the middle of a page) are two registrations of exactly
    # their bytes: the runtime accepts that (profiles/r03_host_register_probe.log), and nothing else on their pages is captured
    back = _own_pages(a.m + a.nnz_jac + 8)
    assert (back.ctypes.data + 8 * a.m) % 4096 != 0
    g2, v2 = back[:a.m], back[a.m:a.m + a.nnz_jac]
    for _ in range(2):
        b.eval_pair(xbuf, g2, v2)
        assert np.array_equal(g2, ref_g) and np.array_equal(v2, ref_v)
    assert b.get_option("pin_held") == 3 and b.get_option("pin_live") == 3
    # a view that overlaps an array the engine holds: ONE region covers both from then on
    merged = b.get_option("pin_merged")
    v2b = back[a.m + 8:a.m + 8 + a.nnz_jac]
    b.eval_pair(xbuf, g2, v2b)
    assert np.array_equal(v2b, ref_v)
    assert b.get_option("pin_merged") == merged + 1 and b.get_option("pin_held") == 4 and b.get_option("pin_live") == 3
    # a request that partly overlaps memory ANOTHER engine holds is refused, counted and reported - and served correctly
    # through the staging buffers
    c = NLPEngine(prob, device=0)
    c.set_option("pin_host", 1)
    refused = c.get_option("pin_overlap_refused")
    big = _own_pages(2 * a.nnz_jac)
    v3 = big[:a.nnz_jac]
    b.eval_pair(xbuf, g2, v3)                                   # b registers the first half of `big`
    v4 = big[a.nnz_jac // 2: a.nnz_jac // 2 + a.nnz_jac]       # c asks for a range that starts inside it and ends beyond
    g4, x4 = _own_pages(a.m), _own_pages(a.n)
    x4[:] = xbuf
    c.eval_pair(x4, g4, v4)
    assert np.array_equal(g4, ref_g) and np.array_equal(v4, ref_v)
    assert c.get_option("pin_overlap_refused") == refused + 1 and "page-lock registry" in c.last_error()
    # arrays below 64 KB are never registered
    small = NLPEngine(problems.launch(2, 4), device=0)
    small.set_option("pin_host", 1)
    assert 8 * small.nnz_jac < 65536
    xs_, gs_, vs_ = small.get_starting_point(), np.zeros(small.m), np.zeros(small.nnz_jac)
    small.eval_pair(xs_, gs_, vs_)
    assert small.get_option("pin_held") == 0
    for e in (small, c, b, ref):
        e.close()


def test_const_once_does_not_trust_a_reallocated_array(built):
    """"const_once" skips the constant tail only if the array is the one it filled last AND sampled tail entries still
    hold what it stored: an array freed and re-allocated at the same address (different contents) gets everything."""
    prob = problems.launch(64, 16)
    ref = NLPEngine(prob, device=0)
    for pin in (0, 1):
        e = NLPEngine(prob, device=0)
        e.set_option("pin_host", pin)
        e.set_option("const_once", 1)
        xs = _iterates(e, 2)
        xbuf, v = np.zeros(e.n), np.zeros(e.nnz_jac)
        xbuf[:] = xs[0]
        e.eval_jac_g(xbuf, True, out=v)
        v[:] = 7.0                                 # same address, the contents of a fresh allocation
        xbuf[:] = xs[1]
        e.eval_jac_g(xbuf, True, out=v)
        assert np.array_equal(v, ref.eval_jac_g(xs[1]))
        e.close()
    ref.close()


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

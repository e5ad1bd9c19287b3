"""Timing of the host-pointer (Ipopt-facing) entry points, PCIe-inclusive, on caller-owned numpy arrays handed again and
again (what Ipopt's TNLPAdapter does with its x / g / jac_g_ arrays).  Used by bench.py (`host_pointer` section) and
tools/host_path_overhead.py.  Wall clock around the calls; every call ends with the engine's own synchronisation."""
import time

import numpy as np

VARIANTS = [
    # name, options, call pattern
    ("two_calls_default_staged", {"pin_host": 0}, "two"),      # the C ABI's default: CPU copies through the engine's page-locked staging buffers
    ("two_calls_pinned", {"pin_host": 1}, "two"),
    ("two_calls_pinned_const_once", {"pin_host": 1, "const_once": 1}, "two"),
    ("two_calls_pinned_delta", {"pin_host": 1, "delta_values": 1}, "two"),
    ("pair_call_pinned", {"pin_host": 1}, "pair"),
    ("pair_call_pinned_delta", {"pin_host": 1, "delta_values": 1}, "pair"),
]


def time_host_path(make_engine, xs, seconds=0.4, variants=None):
    """make_engine() -> a fresh NLPEngine on this rank's device; xs: list of distinct iterates (numpy).
    Returns {variant: {"pairs_per_s", "us_per_pair", "p10_us", "p90_us", "bytes_per_pair"...}}."""
    out = {}
    for name, opts, pattern in VARIANTS:
        if variants is not None and name not in variants:
            continue
        e = make_engine()
        for k, v in opts.items():
            e.set_option(k, v)
        B = e.n_instances
        # four caller-owned x arrays holding distinct iterates, cycled (new_x = true every time; with g and values that is
        # six page-locked registrations, inside the engine's limit of eight); filling them is the caller's work, not timed
        xb = [np.array(xs[i % len(xs)], dtype=np.float64).ravel().copy() for i in range(4)]
        g, v = np.zeros(B * e.m), np.zeros(B * e.nnz_jac)

        def one(i):
            x = xb[i & 3]
            if pattern == "pair":
                e.eval_pair(x, g, v)
            else:
                e.eval_g(x, True, out=g)
                e.eval_jac_g(x, False, out=v)

        for i in range(10):
            one(i)
        ts, i = [], 0
        t_end = time.perf_counter() + seconds
        while time.perf_counter() < t_end or len(ts) < 30:
            t0 = time.perf_counter()
            one(i)
            ts.append(time.perf_counter() - t0)
            i += 1
        ts = np.sort(np.array(ts)) * 1e6
        med = float(np.median(ts))
        rec = {"pairs_per_s": B * 1e6 / med, "us_per_pair": med / B, "p10_us": float(ts[len(ts) // 10]) / B,
               "p90_us": float(ts[len(ts) * 9 // 10]) / B, "calls": len(ts)}
        if opts.get("delta_values"):
            rec["runs_sent"] = e.get_option("delta_sent_runs")
            rec["runs_total"] = e.get_option("delta_total_runs")
        out[name] = rec
        e.close()
    return out


def time_ipopt_iteration(make_engine, xs, seconds=0.3, options=None):
    """One synthetic Ipopt iteration through the host-pointer entry points at ONE iterate: eval_f, eval_grad_f, eval_g,
    eval_jac_g (Core/LpopcIpopt.cpp:106-181), caller-owned arrays; returns median milliseconds."""
    e = make_engine()
    for k, v in (options or {"pin_host": 1}).items():
        e.set_option(k, v)
    xb = [np.array(xs[i % len(xs)], dtype=np.float64).ravel().copy() for i in range(4)]
    g, v = np.zeros(e.n_instances * e.m), np.zeros(e.n_instances * e.nnz_jac)

    def one(i):
        x = xb[i & 3]
        e.eval_f(x, True)
        e.eval_grad_f(x, False)
        e.eval_g(x, False, out=g)
        e.eval_jac_g(x, False, out=v)

    for i in range(5):
        one(i)
    ts, i = [], 0
    t_end = time.perf_counter() + seconds
    while time.perf_counter() < t_end or len(ts) < 20:
        t0 = time.perf_counter()
        one(i)
        ts.append(time.perf_counter() - t0)
        i += 1
    e.close()
    return float(np.median(ts)) * 1e3

#!/usr/bin/env python3
"""bench.py — eval_g+eval_jac_g throughput on the 4-phase, 4096-LGR-point Delta-III problem (BASELINE.json's metric).

A "step" is one fused (eval_g, eval_jac_g) launch over one batch of B distinct NLP iterates x_k (new_x = true for each)
already resident in HBM; R iterates are kept resident and cycled so that the outputs do not live in the Infinity Cache.
The K steps are captured in one hipGraph.  ONE timed region = exactly K steps bracketed by barrier + synchronize; the
region is repeated until at least 0.25 s have been timed and `value` / `ms_per_step` come from the MEDIAN region (p10 / p90
beside it) — a single 20-step region is 2 ms and scatters by 15 %.  Per-rank times are reduced with MAX over ranks.

At --gpus N > 1 `value` is the instance-sharded (weak-scaling) rate of that workload: every rank evaluates its own
stream of iterates, no data-path collective.  The SAME invocation also measures, with all N ranks on the data path:
  strong_scaling   mesh intervals of ONE config-3 / config-4 problem sharded over the ranks, ONE packed RCCL all-gather of
                   the g + Jacobian segments per step, pack / all-gather / unpack captured in the step's hipGraph;
  host_consumer    the same sharding with a host-side consumer (Ipopt's position): every rank stores its runs of g / values
                   into one shared page-locked host array over its own PCIe link, no GPU-to-GPU traffic;
  config5          the 1024-instance quadrotor sweep sharded by instance.

Extra first-class sections on rank 0 at N = 1 (what a TNLP caller can consume):
  sequential       B = 1 device-resident pairs/s (a sequential solver loop on the device)
  host_pointer     PCIe-inclusive pairs/s through rpm_eval_g / rpm_eval_jac_g / rpm_eval_pair on caller-owned arrays
  ms_per_ipopt_iter  eval_f + eval_grad_f + eval_g + eval_jac_g at ONE iterate, device-resident and host-pointer
  device_ipm       the metric problem SOLVED on the device (ms per interior-point iteration); the 1024-instance MPC sweep solved on the device
  roofline         HBM roofline of the dominant kernel: algorithmic bytes per launch / launch duration from HIP events
  cpu_baseline     the CPU oracle (a C port of lpopc's algorithm, oracle/) on one host core, bounded sample
"""
import argparse
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
MIN_TIMED_S = 0.25


def d_tile_doubles(eng):
    return sum(eng.phase_tables(p)["d_vals"].size for p in range(eng.n_phases))


def algorithmic_bytes(eng):
    """read x once per callback, write g, write every Jacobian value, read each D tile once."""
    return 8 * (2 * eng.n + eng.m + eng.nnz_jac) + 8 * d_tile_doubles(eng)


def fused_bytes(eng):
    """What the fused pair launch moves: x once."""
    return 8 * (eng.n + eng.m + eng.nnz_jac) + 8 * d_tile_doubles(eng)


def host_cpu():
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"host_cores": os.cpu_count(), "cpu_model": model}


def cpu_baseline(prob, xs, budget_s):
    """The CPU oracle (C port of the reference algorithm, 1 thread) on a bounded sample, in the two cost shapes of SURVEY
    section 8(d): "faithful-cost" (what lpopc itself pays: COO product with column copies, Find(Doffdiag) scans on every call;
    the top-level figures, kind "port") and "fair" (dense D rows, Doffdiag found once; same bits)."""
    from oracle.oracle import Oracle

    def timed(fair, budget):
        orc = Oracle(prob)
        orc.set_cost_shape(fair)
        orc.eval_g(xs[0])
        orc.eval_jac_g(xs[0])
        t0 = time.perf_counter()
        pairs = 0
        while True:
            x = xs[pairs % len(xs)]
            orc.eval_g(x)
            orc.eval_jac_g(x)
            pairs += 1
            el = time.perf_counter() - t0
            if el >= budget and pairs >= 20:
                break
        return pairs, el

    pairs, el = timed(0, 0.6 * budget_s)
    fpairs, fel = timed(1, 0.4 * budget_s)
    out = {"value": pairs / el, "unit": "pairs/s", "cores": 1, "kind": "port",
           "sample": "%d (eval_g,eval_jac_g) pairs of the same workload in %.1f s, oracle/liborpm.so -O2 -ffp-contract=off, 1 thread "
                     "(the reference is single-threaded), faithful-cost shape" % (pairs, el),
           "fair": {"value": fpairs / fel, "unit": "pairs/s", "cores": 1,
                    "sample": "%d pairs in %.1f s, same library, fair shape: dense per-interval D rows, no per-call Find(Doffdiag) "
                              "scans; bit-identical results" % (fpairs, fel)}}
    out.update(host_cpu())
    return out


def store_ceiling(ctx):
    """What this box's HBM write path delivers to the Jacobian's store pattern, measured in this run (the same binary scatters
    0.55-0.69 of the 8 TB/s peak from box to box): tools/ubench/store_pattern.hip, 256-thread workgroups writing 512-byte
    runs at the stride of a Jacobian block of the metric problem (N = 1024 nodes per phase -> 8 KB), 16 buffers of 107 MB =
    1.7 GB cycling so that nothing lives in the 256 MiB Infinity Cache; a linear 16-B-per-lane fill beside it."""
    import ctypes as C
    torch = ctx.torch
    so = os.path.join(ROOT, "tools", "ubench", "libstore_pattern.so")
    L = C.CDLL(so)
    L.run.restype = C.c_float
    L.run.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_void_p]
    N, instances, nblk, nbuf = 1024, 64, 204, 16
    stride = nblk * N + 16
    bufs = [torch.empty(instances * stride + 64, dtype=torch.float64, device="cuda") for _ in range(nbuf)]
    ptrs = (C.c_void_p * nbuf)(*[b.data_ptr() for b in bufs])
    torch.cuda.synchronize()
    nbytes = instances * nblk * N * 8
    res = {}
    for mode, name in ((1, "pattern_512B_runs"), (2, "linear_16B_per_lane")):
        best = 0.0
        for _ in range(3):
            us = L.run(mode, 0, ptrs, nbuf, N, nblk, instances, stride, 64, None)
            best = max(best, nbytes / us / 1e3)
        res[name + "_GBs"] = best
    res["working_set_GB"] = nbuf * nbytes / 1e9
    res["how"] = "tools/ubench/store_pattern.hip in this process, best of 3 x 64 launches of %.0f MB each" % (nbytes / 1e6)
    del bufs
    torch.cuda.empty_cache()
    return res


class Ctx:
    """Process-wide handles: rank / world, torch, torch.distributed (or None)."""

    def __init__(self, args):
        import torch
        self.torch = torch
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        if not torch.cuda.is_available():
            raise SystemExit("bench.py rank %d of %d: no GPU visible (the hot path has no CPU fallback; needs an MI355X)" % (self.rank, self.world))
        self.dist = None
        torch.cuda.set_device(self.local_rank)
        if self.world > 1 or "RANK" in os.environ:   # under torch.distributed.run: always go through the RCCL path
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("nccl", rank=self.rank, world_size=self.world,
                                    device_id=torch.device("cuda", self.local_rank))
            self.dist = dist

    def sync(self):
        self.torch.cuda.synchronize()

    def barrier_sync(self):
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def max_over_ranks(self, values):
        if self.dist is None:
            return list(values)
        t = self.torch.tensor(list(values), dtype=self.torch.float64, device="cuda")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return [float(v) for v in t.cpu()]


def timed_regions(ctx, step, K, warmup, use_graph=True, min_s=None, max_regions=4000):
    """W warm-up steps, then regions of EXACTLY K steps each (one hipGraph replay when the steps can be captured), each
    bracketed by barrier + synchronize, repeated until min_s seconds have been timed.  Returns wall seconds and device
    milliseconds per region (MAX over ranks), and how the steps were launched."""
    torch = ctx.torch
    min_s = MIN_TIMED_S if min_s is None else min_s
    for k in range(warmup):
        step(k)
    ctx.barrier_sync()
    graph, how = None, "eager launches"
    if use_graph:
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                step(0)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for k in range(K):
                    step(k)
            graph.replay()  # one untimed replay (graph upload)
            how = "hipGraph replay of the K steps"
        except Exception as ex:   # e.g. a collective that cannot be captured: measure eagerly and say so
            graph = None
            how = "eager launches (graph capture failed: %s)" % str(ex).splitlines()[0][:120]
            try:
                torch.cuda.synchronize()
            except Exception:
                pass
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def region():
        ctx.barrier_sync()
        t0 = time.perf_counter()
        ev0.record()
        if graph is not None:
            graph.replay()
        else:
            for k in range(K):
                step(k)
        ev1.record()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        return el, ev0.elapsed_time(ev1)

    first = region()
    second = region()
    # the first region runs slower than the steady state (it set the count too low for min_s in round 1's first cut): the count
    # comes from the faster of the first two, with a margin
    n = int(min(max_regions, max(5, math.ceil(1.15 * min_s / max(min(first[0], second[0]), 1e-6)))))
    n = int(ctx.max_over_ranks([float(n)])[0])   # the same count on every rank
    wall, dev = [first[0], second[0]], [first[1], second[1]]
    for _ in range(n - 2):
        a, b = region()
        wall.append(a)
        dev.append(b)
    ctx.barrier_sync()
    wall = ctx.max_over_ranks(wall)
    dev = ctx.max_over_ranks(dev)
    return {"wall_s": wall, "dev_ms": dev, "launch": how, "graph": graph}


def pct(sorted_vals, q):
    return sorted_vals[min(len(sorted_vals) - 1, int(q * len(sorted_vals)))]


def summarize(regions, K, units_per_step):
    w = sorted(regions["wall_s"])
    d = sorted(regions["dev_ms"])
    med = w[len(w) // 2]
    return {"pairs_per_s": K * units_per_step / med, "ms_per_step": med * 1e3 / K,
            "ms_per_step_p10": pct(w, 0.1) * 1e3 / K, "ms_per_step_p90": pct(w, 0.9) * 1e3 / K,
            "dev_ms_per_step": d[len(d) // 2] / K, "timed_regions": len(w), "timed_seconds": sum(w),
            "first_region_ms_per_step": regions["wall_s"][0] * 1e3 / K, "launch": regions["launch"]}


def make_iterates(problems, eng, count, seed0, mode="perturb"):
    xl, xu, _, _ = eng.get_bounds_info()
    x0 = eng.get_starting_point()
    return [problems.seeded_iterate(x0, xl, xu, seed0 + r, mode) for r in range(count)]


def device_workload(ctx, args, prob, B, R, K, warmup, sharded, seed0, mode="perturb", align=16, use_graph=True,
                    role_loop=-1, pipeline=-1, dx_mode=0, tile_nodes=0, unfused=False, keep=False, persistent=False):
    """The fused pair kernel over R resident iterates, B per step; with `sharded` the mesh intervals of every iterate are split
    over the ranks and the results exchanged with ONE packed all-gather per step."""
    import numpy as np
    from lpopc_amd import problems
    from lpopc_amd.engine import NLPEngine
    torch = ctx.torch
    sh = sharded and ctx.world > 1
    eng = NLPEngine(prob, n_instances=B, shard_mode=1 if sh else 0, shard_rank=ctx.rank if sh else 0,
                    shard_world=ctx.world if sh else 1, tile_nodes=tile_nodes, device=ctx.local_rank, role_loop=role_loop)
    if dx_mode:
        eng.set_option("dx_mode", dx_mode)
    if pipeline != -1:
        eng.set_option("pipeline", pipeline)
    eng.set_option("instance_align", align)   # every iterate's g / values array starts on a 128-byte line
    if persistent:
        eng.set_option("persistent_values", 1)  # the constant Doffdiag block of each resident values array is written once
    R = max(R, 2 * B)
    R -= R % B
    xs = make_iterates(problems, eng, R, seed0, mode)
    d_x = torch.from_numpy(np.stack(xs)).cuda()
    sg, sv = eng.get_option("stride_g"), eng.get_option("stride_values")
    d_g = torch.empty((R, sg), dtype=torch.float64, device="cuda")
    d_v = torch.empty((R, sv), dtype=torch.float64, device="cuda")
    xch = None
    if sh:
        from lpopc_amd.dist import IntervalExchange
        xch = IntervalExchange(eng, ctx.dist, ctx.world, ctx.rank)

    def step(k):
        r = (k * B) % R          # this step's batch of B consecutive resident iterates
        if unfused:
            eng.eval_g_dev(d_x[r], d_g[r])
            eng.eval_jac_g_dev(d_x[r], d_v[r])
        else:
            eng.eval_pair_dev(d_x[r], d_g[r], d_v[r])
        if xch is not None:
            xch.exchange(d_g[r], d_v[r])

    regions = timed_regions(ctx, step, K, warmup, use_graph=use_graph)
    regions.pop("graph", None)
    out = summarize(regions, K, B)
    out.update({"iterates_per_step": B, "resident_iterates": R, "n": eng.n, "m": eng.m, "nnz_jac": eng.nnz_jac,
                "kernel": ("rpm_tile_pl_kernel" if eng.get_option("pipeline_active") else
                           "rpm_tile_rl_kernel" if eng.get_option("role_loop") else "rpm_tile_kernel"),
                "tile_nodes": eng.get_option("tile_nodes")})
    if xch is not None:
        out["allgather_bytes_received_per_step"] = xch.bytes_received_per_step()
        out["collective"] = "one in-place RCCL all_gather_into_tensor of the packed [world][slot] buffer per step"
    if keep:
        return out, eng, xs, d_x, d_g, d_v
    eng.close()
    return out


def check_against_single_evaluation(ctx, args, prob, eng, d_x, d_g, d_v, R):
    """Results of the timed region are finite and equal to a fresh single evaluation."""
    from lpopc_amd.engine import NLPEngine
    torch = ctx.torch
    one = NLPEngine(prob, tile_nodes=args.tile_nodes, device=ctx.local_rank)
    if args.dx_mode:
        one.set_option("dx_mode", args.dx_mode)
    chk_g = torch.empty(eng.m, dtype=torch.float64, device="cuda")
    chk_v = torch.empty(eng.nnz_jac, dtype=torch.float64, device="cuda")
    for r in (0, R - 1):
        one.eval_pair_dev(d_x[r], chk_g, chk_v)
        torch.cuda.synchronize()
        assert torch.equal(chk_g, d_g[r, :eng.m]) and torch.equal(chk_v, d_v[r, :eng.nnz_jac]) and bool(torch.isfinite(chk_v).all())
    one.close()


def sequential_section(ctx, args, prob, xs):
    """B = 1: what a sequential solver loop on the device sees (latency-bound), and one synthetic Ipopt iteration
    (eval_f + eval_grad_f + eval_g + eval_jac_g at ONE iterate) with everything resident in HBM."""
    import numpy as np
    from lpopc_amd.engine import NLPEngine
    torch = ctx.torch
    one = NLPEngine(prob, tile_nodes=args.tile_nodes, device=ctx.local_rank)
    R1 = min(len(xs), 256)
    d_x = torch.from_numpy(np.stack(xs[:R1])).cuda()
    d_g = torch.empty((R1, one.m), dtype=torch.float64, device="cuda")
    d_v = torch.empty((R1, one.nnz_jac), dtype=torch.float64, device="cuda")
    d_f = torch.empty(1, dtype=torch.float64, device="cuda")
    d_grad = torch.empty((R1, one.n), dtype=torch.float64, device="cuda")
    K1 = 256
    reg = timed_regions(ctx, lambda k: one.eval_pair_dev(d_x[k % R1], d_g[k % R1], d_v[k % R1]), K1, 20)
    reg.pop("graph", None)
    s = summarize(reg, K1, 1)
    us = s["dev_ms_per_step"] * 1e3
    seq = {"pairs_per_s": s["pairs_per_s"], "us_per_pair": s["ms_per_step"] * 1e3, "launch_us": us,
           "hbm_frac": algorithmic_bytes(one) / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
           "hbm_frac_fused_bytes": fused_bytes(one) / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
           "kernel": "rpm_tile_kernel (one role per thread, one iterate per launch)", "timed_regions": s["timed_regions"],
           "launch": s["launch"]}

    def it(k):
        r = k % R1
        one.eval_f_dev(d_x[r], d_f)
        one.eval_grad_f_dev(d_x[r], d_grad[r])
        one.eval_pair_dev(d_x[r], d_g[r], d_v[r])
    reg = timed_regions(ctx, it, 64, 10)
    reg.pop("graph", None)
    s2 = summarize(reg, 64, 1)
    one.close()
    return seq, s2["ms_per_step"]



def _perturbed_starts(ipm, x0, seeds=(1, 2, 3, 4, 5)):
    """The metric problem from starts perturbed by 1e-10 (relative; tools/ipm_delta3.py's IPM_PERTURB_SEED): how many end at the
    optimum (status 0 or the acceptable level, within 0.05 kg of the published 7529.71), median time and iteration count."""
    import numpy as np
    runs = []
    for seed in seeds:
        xs = x0 * (1 + 1e-10 * np.random.RandomState(seed).uniform(-1, 1, x0.shape))
        t0 = time.perf_counter()
        r = ipm.solve(xs)
        runs.append((int(r["status"][0]), time.perf_counter() - t0, -float(r["obj"][0]) * 301454.0, int(r["iterations"][0])))
    good = [q for q in runs if q[0] in (0, 1) and abs(q[2] - 7529.71) < 0.05]
    return {"starts": len(runs), "at_the_optimum": len(good), "median_solve_s": float(np.median([q[1] for q in runs])),
            "median_iterations": int(np.median([q[3] for q in runs])), "statuses": [q[0] for q in runs]}

def device_ipm_section(ctx, args):
    """Row f-2 beside the callbacks: the METRIC problem solved on the device from lpopc's default guess (callbacks, exact
    Hessian, KKT assembly, nested-dissection LDL^T, substitution, filter line search with second-order corrections,
    restoration phase), wall time per interior-point iteration; and the 1024-instance quadrotor MPC sweep solved to 1e-8."""
    import numpy as np
    from lpopc_amd import problems
    from lpopc_amd.engine import BatchedIPM, NLPEngine
    from lpopc_amd.problem import Options
    o = Options()
    o.SetStringValue("hessian-approximation", "exact")
    out = {}
    eng = NLPEngine(problems.launch(args.intervals, args.nodes), o, device=ctx.local_rank)
    ipm = BatchedIPM(eng, max_iter=2000)
    x0 = eng.get_starting_point()[None, :]
    ipm.set_option("mu_strategy", "monotone")           # the other barrier rule first, for the record
    t0 = time.perf_counter()
    rm = ipm.solve(x0)
    dtm = time.perf_counter() - t0
    stm = ipm.stats()
    monotone = {"solve_s": dtm, "status": int(rm["status"][0]), "iterations": int(rm["iterations"][0]),
                "ms_per_ipm_iteration": 1e3 * dtm / max(1, stm["iterations"]), "final_mass_kg": -float(rm["obj"][0]) * 301454.0}
    ipm.set_option("mu_strategy", "adaptive")           # the default (what the reference asks Ipopt for): the figures below
    t0 = time.perf_counter()
    r = ipm.solve(x0)
    dt = time.perf_counter() - t0
    st, info, kt = ipm.stats(), ipm.info(), ipm.kernel_times()
    out["metric_problem"] = {"solve_s": dt, "status": int(r["status"][0]), "iterations": int(r["iterations"][0]),
                             "ms_per_ipm_iteration": 1e3 * dt / max(1, st["iterations"]), "objective": float(r["obj"][0]),
                             "final_mass_kg": -float(r["obj"][0]) * 301454.0, "kkt_error": float(r["kkt_error"][0]),
                             "restorations": int(ipm.restorations()[0]), "factorizations": st["factorizations"], "trial_points": st["trial_points"],
                             "factor_ms_per_launch": kt["factor_ms"] / max(1, st["factorizations"]),
                             "kkt_order": info["kkt_order"], "sub_problems": int(ipm.subproblems().shape[0]), "mu_strategy": "adaptive",
                             "with_mu_strategy_monotone": monotone,
                             "note": "Delta-III %dx%dx%d from lpopc's default guess; status 0 converged (1e-8), 1 acceptable level; published optimum 7529.71 kg" % (
                                 4, args.intervals, args.nodes)}
    # the path is chaotic (a start perturbed by 1e-10 takes another one, 300 ... 900 iterations): the same solve from five such starts
    out["metric_problem"]["perturbed_starts"] = _perturbed_starts(ipm, x0)
    try:   # Ipopt's own default NLP scaling (gradient-based; an option here): 28 672 of the 32 801 rows are scaled down
        ipm.set_option("nlp_scaling", 1)
        t0 = time.perf_counter()
        rs = ipm.solve(x0)
        dts = time.perf_counter() - t0
        sts = ipm.stats()
        out["metric_problem"]["with_nlp_scaling_gradient_based"] = {
            "solve_s": dts, "status": int(rs["status"][0]), "iterations": int(rs["iterations"][0]), "factorizations": sts["factorizations"],
            "final_mass_kg": -float(rs["obj"][0]) * 301454.0, "kkt_error": float(rs["kkt_error"][0]),
            "note": "option nlp_scaling = 1 (Ipopt's nlp_scaling_method default; off by default here: ensembles over meshes in DESIGN.md f-2)"}
        ipm.set_option("nlp_scaling", 0)
    except Exception as ex:
        out["metric_problem"]["with_nlp_scaling_gradient_based"] = {"error": repr(ex)}
    ipm.close()
    eng.close()
    # the same problem with lpopc's DEFAULT option hessian-approximation = limited-memory (Core/LpNLPWrapper.hpp:71): Ipopt's
    # limited-memory BFGS on the device (csrc/rpm_ipm_lbfgs.hip), no Hessian evaluation, 12 more substitutions per iteration
    try:
        eng = NLPEngine(problems.launch(args.intervals, args.nodes), device=ctx.local_rank)
        ipm = BatchedIPM(eng, max_iter=3000)
        t0 = time.perf_counter()
        r = ipm.solve(eng.get_starting_point()[None, :])
        dt = time.perf_counter() - t0
        st, info = ipm.stats(), ipm.info()
        out["metric_problem_limited_memory"] = {
            "solve_s": dt, "status": int(r["status"][0]), "iterations": int(r["iterations"][0]), "ms_per_ipm_iteration": 1e3 * dt / max(1, st["iterations"]),
            "final_mass_kg": -float(r["obj"][0]) * 301454.0, "kkt_error": float(r["kkt_error"][0]), "factorizations": st["factorizations"],
            "half_bandwidth": info["half_bandwidth"], "hessian": "limited-memory BFGS, history 6 (lpopc's default option)",
            "perturbed_starts": _perturbed_starts(ipm, eng.get_starting_point()[None, :])}
        ipm.close()
        eng.close()
    except Exception as ex:
        out["metric_problem_limited_memory"] = {"error": repr(ex)}
    B = 1024
    prob = problems.quadrotor(8, 8)
    eng = NLPEngine(prob, o, n_instances=B, device=ctx.local_rank)
    eng.set_option("instance_align", 16)
    ipm = BatchedIPM(eng)
    xl, xu, _, _ = NLPEngine(prob, o).get_bounds_info()
    x_start = eng.get_starting_point()[:eng.n]
    rng = np.random.RandomState(5)
    idx = [i * 65 for i in range(12)]
    L, U = np.tile(xl, (B, 1)), np.tile(xu, (B, 1))
    for bi in range(B):
        L[bi, idx] = U[bi, idx] = np.concatenate([rng.uniform(-0.5, 0.5, 3), rng.uniform(-0.3, 0.3, 3), rng.uniform(-0.1, 0.1, 6)])
    ipm.set_all_bounds(L, U)
    x0 = np.tile(x_start, (B, 1))
    sweep = {"note": "host arrays in and out (rpm_ipm_solve), per-instance initial states; mu_strategy adaptive is the solver's default (what "
                     "the reference asks Ipopt for), monotone needs fewer iterations on this well-behaved sweep"}
    for name, code in (("adaptive", 1), ("monotone", 0)):
        ipm.set_option("mu_strategy", code)
        ipm.solve(x0)
        t0 = time.perf_counter()
        r = ipm.solve(x0)
        dt = time.perf_counter() - t0
        sweep[name] = {"solve_s": dt, "solves_per_s": B / dt, "converged": int((r["status"] == 0).sum()),
                       "batched_iterations": ipm.stats()["iterations"], "max_kkt_error": float(r["kkt_error"].max())}
    sweep.update(sweep["adaptive"])        # the default's figures at the top level
    out["config5_sweep_1024"] = sweep
    ipm.close()
    eng.close()
    return out


def host_section(ctx, args, prob, xs):
    from lpopc_amd.engine import NLPEngine
    from lpopc_amd.hostbench import time_host_path, time_ipopt_iteration
    mk = lambda: NLPEngine(prob, device=ctx.local_rank)   # noqa: E731
    res = time_host_path(mk, xs[:4], seconds=0.4)
    it_plain = time_ipopt_iteration(mk, xs[:4], options={"pin_host": 1})
    it_delta = time_ipopt_iteration(mk, xs[:4], options={"pin_host": 1, "delta_values": 1})
    note = ("PCIe-inclusive, wall clock around the C-ABI calls on caller-owned arrays handed again and again (as Ipopt's "
            "TNLPAdapter does); never `value`.  two_calls = rpm_eval_g(new_x=1) + rpm_eval_jac_g(new_x=0); pair_call = "
            "rpm_eval_pair; const_once / delta = only the part of `values` that changed crosses PCIe (rpm_hip.h)")
    return {"note": note, "variants": res}, it_plain, it_delta


def host_consumer_section(ctx, args, prob, B=1):
    """All ranks store their interval shares of g / values into ONE shared page-locked host array (no GPU-to-GPU traffic)."""
    import numpy as np
    from lpopc_amd import problems
    from lpopc_amd.dist import HostConsumerGroup
    from lpopc_amd.engine import NLPEngine
    sh = ctx.world > 1
    eng = NLPEngine(prob, n_instances=B, shard_mode=1 if sh else 0, shard_rank=ctx.rank if sh else 0,
                    shard_world=ctx.world if sh else 1, device=ctx.local_rank)
    grp = None
    try:
        if ctx.dist is None:
            raise RuntimeError("needs torch.distributed")
        grp = HostConsumerGroup(eng, ctx.dist, eng.n, eng.m, eng.nnz_jac, n_instances=B)
        xs = make_iterates(problems, eng, 4 * B, 3)
        if ctx.rank == 0:
            for i in range(4):
                grp.x[i][:] = np.concatenate(xs[i * B:(i + 1) * B])
        ctx.dist.barrier()
        for k in range(10):
            grp.step(k & 3)
        K = 200
        ts = []
        for rep in range(6):
            ctx.dist.barrier()
            t0 = time.perf_counter()
            for k in range(K):
                grp.step(k & 3)
            ts.append(time.perf_counter() - t0)
        ts = sorted(ctx.max_over_ranks(ts))
        med = ts[len(ts) // 2]
        ok = True
        if ctx.rank == 0:   # the assembled arrays equal a single-GPU evaluation of the last iterate
            ref = NLPEngine(prob, n_instances=B, device=ctx.local_rank)
            g_ref, v_ref = ref.eval_pair(np.array(grp.x[(K - 1) & 3]))
            ok = bool(np.array_equal(g_ref, np.array(grp.g)) and np.array_equal(v_ref, np.array(grp.values)))
            ref.close()
        return {"pairs_per_s": K * B / med, "us_per_pair": med * 1e6 / (K * B), "iterates_per_call": B,
                "equals_single_gpu_result": ok, "timed_regions": len(ts),
                "how": "interval-sharded engines, x read from / g stored into / changed runs of values stored into one shared "
                       "page-locked host segment, every rank over its own PCIe link; go/done words in the same segment"}
    finally:
        eng.close()          # releases its page-locked registrations of the segment before the segment is unmapped
        if grp is not None:
            grp.close()


def group_section(ctx, args, prob, devices):
    """rpm_group_*: ONE process (this rank) drives every listed device — the configuration lpopc's single-process
    NLPSolver::SolveNlp (Core/LpNLPSolver.cpp:13-53) can use.  Mesh intervals of the metric problem sharded over the devices, one
    iterate per call; wall clock around the C-ABI calls (blocking), results checked against a single engine."""
    import mmap
    import numpy as np
    from lpopc_amd import problems
    from lpopc_amd.engine import NLPEngine
    from lpopc_amd.group import EngineGroup
    torch = ctx.torch
    own = lambda n: np.frombuffer(mmap.mmap(-1, 8 * n), dtype=np.float64, count=n)   # noqa: E731
    grp = EngineGroup(prob, devices)
    one = NLPEngine(prob, device=devices[0])
    out = {"devices": list(devices), "what": "one process, one interval-sharded engine per device (rpm_group_*), metric problem, one "
                                             "iterate per call, wall clock around the blocking calls"}
    try:
        grp.device_init()
        xs = make_iterates(problems, one, 4, 3)
        xb = [own(one.n) for _ in range(4)]
        for i in range(4):
            xb[i][:] = xs[i]
        g, v = own(one.m), own(one.nnz_jac)

        def timed(call, K=200, reps=5):
            for k in range(10):
                call(k)
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                for k in range(K):
                    call(k)
                ts.append((time.perf_counter() - t0) / K)
            return sorted(ts)[len(ts) // 2] * 1e6

        us = timed(lambda k: grp.eval_pair(xb[k & 3], g, v))
        ref_g, ref_v = one.eval_pair(xs[3])
        ok = bool(np.array_equal(g, ref_g) and np.array_equal(v, ref_v))
        out["host_consumer_pair_call"] = {"us_per_pair": us, "pairs_per_s": 1e6 / us, "equals_single_engine": ok,
                                          "how": "rpm_group_eval_pair: every device reads x from and stores its rows of g / changed runs of "
                                                 "values into the caller's page-locked arrays over its own PCIe link"}

        def two(k):
            grp.eval_g(xb[k & 3], g, True)
            grp.eval_jac_g(xb[k & 3], v, False)
        us2 = timed(two)
        out["host_consumer_two_calls"] = {"us_per_pair": us2, "pairs_per_s": 1e6 / us2,
                                          "how": "rpm_group_eval_g(new_x = 1) + rpm_group_eval_jac_g(new_x = 0), as Ipopt calls them"}
        # device consumer: arrays on the first device, the others store into them over xGMI; then the one-shot all-gather
        d0 = torch.device("cuda", devices[0])
        d_x = [torch.from_numpy(x).to(d0) for x in xs]
        d_g = torch.empty(one.m, dtype=torch.float64, device=d0)
        d_v = torch.empty(one.nnz_jac, dtype=torch.float64, device=d0)
        rg = torch.empty(one.m, dtype=torch.float64, device=d0)
        rv = torch.empty(one.nnz_jac, dtype=torch.float64, device=d0)
        for dv in set(devices):
            torch.cuda.synchronize(dv)
        us3 = timed(lambda k: grp.eval_pair_dev(0, d_x[k & 3], d_g, d_v))
        with torch.cuda.device(d0):
            one.eval_pair_dev(d_x[3], rg, rv)
            torch.cuda.synchronize()
        out["device_consumer_direct_peer_stores"] = {"us_per_pair": us3, "pairs_per_s": 1e6 / us3,
                                                     "equals_single_engine": bool(torch.equal(d_g, rg) and torch.equal(d_v, rv)),
                                                     "how": "rpm_group_eval_pair_dev: x, g, values in the first device's HBM; the other devices' "
                                                            "tile kernels read and store them over xGMI (peer access), no pack / gather"}
        ax, ag, av = [], [], []
        for r, dv in enumerate(devices):
            dd = torch.device("cuda", dv)
            ax.append([torch.from_numpy(x).to(dd) for x in xs])
            ag.append(torch.empty(one.m, dtype=torch.float64, device=dd))
            av.append(torch.empty(one.nnz_jac, dtype=torch.float64, device=dd))
        for dv in set(devices):
            torch.cuda.synchronize(dv)
        us4 = timed(lambda k: grp.allgather_pair_dev([a[k & 3] for a in ax], ag, av))
        last = len(devices) - 1
        out["device_consumer_allgather_peer_push"] = {"us_per_pair": us4, "pairs_per_s": 1e6 / us4,
                                                      "equals_single_engine": bool(torch.equal(ag[last].to(d0), rg) and torch.equal(av[last].to(d0), rv)),
                                                      "how": "rpm_group_allgather_pair_dev: every device fills its share of its own arrays and ONE push "
                                                             "kernel per device stores it into every peer's, one xGMI link per peer"}
    finally:
        grp.close()
        one.close()
        torch.cuda.set_device(ctx.local_rank)
    try:   # rpm_sweep_*: the device solver's 1024-instance sweep dealt to the same devices (instances, not intervals: nothing crosses)
        out["sweep_solver_instances_dealt"] = sweep_group_part(devices)
    except Exception as ex:
        out["sweep_solver_instances_dealt"] = {"error": repr(ex)}
    torch.cuda.set_device(ctx.local_rank)
    return out


def sweep_group_part(devices, B=1024):
    import numpy as np
    from lpopc_amd import problems
    from lpopc_amd.group import SweepGroup
    from lpopc_amd.problem import Options
    o = Options()
    o.SetStringValue("hessian-approximation", "exact")
    prob = problems.quadrotor(8, 8)
    sw = SweepGroup(prob, devices, B, o)
    try:
        from lpopc_amd.engine import NLPEngine
        one = NLPEngine(prob, o)
        xl, xu, _, _ = one.get_bounds_info()
        x_start = one.get_starting_point()[:one.n]
        one.close()
        rng = np.random.RandomState(5)
        idx = [i * 65 for i in range(12)]
        for bi in range(B):
            l, u = xl.copy(), xu.copy()
            l[idx] = u[idx] = np.concatenate([rng.uniform(-0.5, 0.5, 3), rng.uniform(-0.3, 0.3, 3), rng.uniform(-0.1, 0.1, 6)])
            sw.set_bounds(bi, l, u)
        x0 = np.tile(x_start, (B, 1))
        sw.solve(x0)
        t0 = time.perf_counter()
        r = sw.solve(x0)
        dt = time.perf_counter() - t0
        return {"devices": list(devices), "instances": B, "solve_s": dt, "solves_per_s": B / dt, "converged": int((r["status"] == 0).sum()),
                "max_kkt_error": float(r["kkt_error"].max()), "batched_iterations": sw.stats()["iterations"],
                "how": "rpm_sweep_solve: one process, an engine and a solver per device, a host thread per share; host arrays in and out"}
    finally:
        sw.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--iterates", type=int, default=384,
                    help="distinct NLP iterates resident in HBM: 384 x 7.1 MB = 2.7 GB of outputs, ten times the 256 MiB Infinity "
                         "Cache (with 64 resident iterates and 16 per launch the Jacobian blocks of a cycle fit it and the kernel "
                         "runs 13 %% faster than HBM allows)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--unfused", action="store_true", help="separate eval_g and eval_jac_g kernels per step")
    ap.add_argument("--shard", choices=["instances", "intervals"], default="instances",
                    help="what `value` measures at --gpus N > 1 (the other split is reported in its own section either way)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--only-main", action="store_true", help="skip every section but the main metric")
    ap.add_argument("--profile", action="store_true",
                    help="profiling run: only the warm-up and ONE timed region (no extra checks/sections), so that\n"
                         "rocprofv3 --stats averages exactly the launches bench.py times")
    ap.add_argument("--dx-mode", type=int, default=0, help="0: scalar D.X in the reference's order, 1: FP64 MFMA tiles")
    ap.add_argument("--tile-nodes", type=int, default=0)
    ap.add_argument("--role-loop", type=int, default=-1, help="-1 auto, 0 one role per thread, 1 role-looped 64-node tiles")
    ap.add_argument("--instance-align", type=int, default=16,
                    help="doubles; start of every instance's g / values array inside a batch (1 = packed back to back)")
    ap.add_argument("--pipeline", type=int, default=-1, help="-1 auto, 0 role-looped kernel only, 1 force the pipelined kernel")
    ap.add_argument("--batch", type=int, default=64,
                    help="NLP iterates evaluated per launch (independent instances of the same problem): 64 amortise the\n"
                         "persistent kernel's prologue and tail (0.68 of the HBM peak; 16 per launch: 0.55-0.56)")
    ap.add_argument("--persistent", action="store_true",
                    help="main workload with option persistent_values (profiling of that mode; the line then says so in config.pair)")
    ap.add_argument("--intervals", type=int, default=64)
    ap.add_argument("--nodes", type=int, default=16)
    ap.add_argument("--group-devices", type=str, default="",
                    help="devices of the single-process group section (rpm_group_*), e.g. 0,0,0,0 to rehearse on one GPU; default: all "
                         "N devices of the run, on rank 0, when N > 1")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # a bare `python bench.py --gpus N`: start the N ranks (one process per GPU) ourselves, before anything touches the
        # GPU in this process, and pass their exit code on
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "1")
        sys.exit(subprocess.call(cmd, env=env))

    from lpopc_amd import problems
    ctx = Ctx(args)
    rank, world = ctx.rank, ctx.world
    global MIN_TIMED_S
    if args.profile:
        MIN_TIMED_S = 0.0
    errors = {}

    prob = problems.launch(args.intervals, args.nodes)
    sharded = args.shard == "intervals" and world > 1
    B = max(1, args.batch)
    # rank-specific iterates in the weak-scaling mode, identical ones when one problem is sharded
    seed0 = 3 if sharded else 3 + 1000 * rank
    main_res, eng, xs, d_x, d_g, d_v = device_workload(
        ctx, args, prob, B, args.iterates, args.steps, args.warmup, sharded, seed0, align=args.instance_align,
        use_graph=not args.no_graph, role_loop=args.role_loop, pipeline=args.pipeline, dx_mode=args.dx_mode,
        tile_nodes=args.tile_nodes, unfused=args.unfused, keep=True, persistent=args.persistent)
    R = main_res["resident_iterates"]
    if not sharded and not os.environ.get("RPM_DIAG_MASK") and not args.profile:
        check_against_single_evaluation(ctx, args, prob, eng, d_x, d_g, d_v, R)

    out = None
    if rank == 0:
        units = B * (1 if sharded else world)
        launches_per_step = 2 if args.unfused else 1
        dev_ms_per_step = main_res["dev_ms_per_step"]
        bytes_per_launch = algorithmic_bytes(eng) * B
        achieved = bytes_per_launch / (dev_ms_per_step * 1e-3) / 1e9   # algorithmic bytes of a step / device time of a step
        achieved_fused = fused_bytes(eng) * B / (dev_ms_per_step * 1e-3) / 1e9
        traffic, traffic_source = None, None
        # HBM bytes per launch from the rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs) of this exact
        # command, committed under profiles/ (tools/collect_profiles.sh regenerates it): STATIC, not measured in this run
        tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tp) and B == 64 and not args.unfused and args.intervals == 64 and args.nodes == 16:   # the default command
            try:
                tj = json.load(open(tp))
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_source = "static: profiles/pmc_traffic.json (%s), rocprofv3 --pmc passes of this command, not measured in this run" % tj.get("tag", "r01")
            except Exception:
                traffic = None
        out = {
            "metric": "eval_g+eval_jac_g calls/sec, 4-phase 4096-LGR-pt problem",
            "value": main_res["pairs_per_s"] * (1 if sharded else world),
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": main_res["ms_per_step"],
            "ms_per_step_p10": main_res["ms_per_step_p10"],
            "ms_per_step_p90": main_res["ms_per_step_p90"],
            "timed_regions": main_res["timed_regions"],
            "timed_seconds": main_res["timed_seconds"],
            "first_region_ms_per_step": main_res["first_region_ms_per_step"],
            "higher_is_better": True,
            "scaling": "strong" if sharded else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "Delta-III 4-phase launch ascent, %d intervals/phase x %d LGR points (n=%d, m=%d, nnz_jac=%d), "
                            "first-derive=finite-difference tol=1e-6, %d seeded iterates resident in HBM, %d iterate(s) per step"
                            % (args.intervals, args.nodes, eng.n, eng.m, eng.nnz_jac, R, B),
                "pair": ("unfused: eval_g kernel + eval_jac_g kernel" if args.unfused else
                         "fused: one tile-kernel launch writes g and all Jacobian values of the step's iterates") +
                        (" EXCEPT the constant Doffdiag block (--persistent: not the TNLP contract, not comparable with `value` of a default run)" if args.persistent else ""),
                "timing": "each timed region = exactly `steps` steps between barrier+synchronize; regions repeated until >= %.2f s; "
                          "value and ms_per_step are the median region (max over ranks per region)" % MIN_TIMED_S,
                "launch": main_res["launch"],
                "tile_nodes": main_res["tile_nodes"],
                "thread_layout": ("persistent workgroups of two halves, each 4 compute waves + 2 DMA waves (rpm_tile_pl_kernel)" if main_res["kernel"] == "rpm_tile_pl_kernel"
                                  else "64 nodes x 4 role groups (roles looped)" if main_res["kernel"] == "rpm_tile_rl_kernel" else "16 nodes x (nx+nu+2) roles"),
                "dx_mode": "mfma_f64_16x16x4" if args.dx_mode else "scalar, reference summation order",
                "parallelism": ("intervals sharded x%d + one packed RCCL all-gather per step" % world) if sharded else
                               ("independent instances x%d" % world),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "frac_fused_bytes": achieved_fused / HBM_PEAK_GBS,
                "traffic": traffic, "traffic_source": traffic_source,
                "kernel": main_res["kernel"],
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "algorithmic_bytes_per_pair": bytes_per_launch // B,
                "fused_bytes_per_pair": fused_bytes(eng),
                "avg_launch_us": dev_ms_per_step * 1e3 / launches_per_step,
                "how": "median over the timed regions of (HIP-event time of the region / launches in it)",
            },
        }
    xs_main = xs
    stored_bytes_per_launch = B * (eng.m + eng.nnz_jac) * 8
    eng.close()
    del d_x, d_g, d_v
    ctx.torch.cuda.empty_cache()
    if rank == 0 and not args.profile and not args.persistent:
        try:   # normalise by what this box's write path delivers to the same store pattern, measured now
            sc = store_ceiling(ctx)
            out["roofline"]["measured_store_ceiling"] = sc
            # like with like: the bytes the kernel STORES per launch (g and the Jacobian values of every resident iterate: what
            # rocprofv3's WRITE_SIZE counts, profiles/pmc_traffic.json) over its launch time, against the pure store stream;
            # `achieved` above also counts the algorithmic reads of SURVEY 8(d), most of which the caches serve
            out["roofline"]["stored_bytes_per_launch"] = stored_bytes_per_launch
            out["roofline"]["store_rate_GBs"] = stored_bytes_per_launch / (out["roofline"]["avg_launch_us"] * 1e-6) / 1e9
            out["roofline"]["frac_of_measured_store_ceiling"] = out["roofline"]["store_rate_GBs"] / sc["pattern_512B_runs_GBs"]
        except Exception as ex:
            errors["store_ceiling"] = repr(ex)

    extras = not args.profile and not args.only_main
    # ---- rank 0 alone, N = 1: what a TNLP caller can consume --------------------------------------------------------
    if extras and world == 1 and rank == 0:
        try:
            seq, it_dev_ms = sequential_section(ctx, args, prob, xs_main)
            out["sequential"] = seq
            out["ms_per_ipopt_iter"] = {"device_resident_b1": it_dev_ms,
                                        "what": "eval_f + eval_grad_f + eval_g + eval_jac_g at ONE iterate (Core/LpopcIpopt.cpp:106-181), "
                                                "median; Ipopt itself is absent, so no linear-solver time is in it"}
        except Exception as ex:
            errors["sequential"] = repr(ex)
        try:   # SURVEY 8(d)'s persistent-buffer variant: same workload, the constant block of every resident array written once
            pv = device_workload(ctx, args, prob, B, args.iterates, args.steps, args.warmup, False, 3, align=args.instance_align,
                                 use_graph=not args.no_graph, role_loop=args.role_loop, pipeline=args.pipeline, tile_nodes=args.tile_nodes,
                                 persistent=True)
            from lpopc_amd.engine import NLPEngine
            e1 = NLPEngine(prob)
            i_, j_ = e1.eval_jac_g_structure()
            nnz_const = int(sum(e1.phase_tables(p)["doff_vals"].size * e1._desc.phases[p].nx for p in range(e1.n_phases)))
            bprime = algorithmic_bytes(e1) - 8 * nnz_const
            e1.close()
            ach = bprime * B / (pv["dev_ms_per_step"] * 1e-3) / 1e9
            out["persistent_values"] = {
                "pairs_per_s": pv["pairs_per_s"], "ms_per_step": pv["ms_per_step"], "dev_ms_per_step": pv["dev_ms_per_step"],
                "algorithmic_bytes_per_pair_Bprime": bprime, "achieved_GBs": ach, "frac": ach / HBM_PEAK_GBS,
                "kernel": pv["kernel"], "iterates_per_step": B,
                "what": "option persistent_values: the %d resident values arrays keep their constant Doffdiag block (%d of %d entries) from the "
                        "first fill; never `value` — the TNLP contract hands a caller buffer and counts every entry" % (pv["resident_iterates"], nnz_const, pv["nnz_jac"])}
            out["roofline"]["persistent_values"] = {"frac": ach / HBM_PEAK_GBS, "achieved": ach, "bytes_per_pair": bprime,
                                                    "avg_launch_us": pv["dev_ms_per_step"] * 1e3, "pairs_per_s": pv["pairs_per_s"]}
        except Exception as ex:
            errors["persistent_values"] = repr(ex)
        try:
            out["device_ipm"] = device_ipm_section(ctx, args)
        except Exception as ex:
            errors["device_ipm"] = repr(ex)
        try:
            hp, it_plain, it_delta = host_section(ctx, args, prob, xs_main)
            out["host_pointer"] = hp
            out.setdefault("ms_per_ipopt_iter", {})["host_pointer_pinned"] = it_plain
            out["ms_per_ipopt_iter"]["host_pointer_pinned_delta"] = it_delta
        except Exception as ex:
            errors["host_pointer"] = repr(ex)
        # the figures a TNLP caller actually consumes, where the driver keeps values (never `value`)
        shaped = {"what": "pairs/s of ONE iterate per call, same binary, same run; `value` above is the batched figure"}
        if "sequential" in out:
            shaped["device_resident_b1_pairs_per_s"] = out["sequential"]["pairs_per_s"]
            out["roofline"]["b1_device_resident"] = {"pairs_per_s": out["sequential"]["pairs_per_s"], "launch_us": out["sequential"]["launch_us"],
                                                     "frac": out["sequential"]["hbm_frac"], "bound": "latency (one 7 us launch per pair)"}
        if "host_pointer" in out:
            for name, r in out["host_pointer"]["variants"].items():
                shaped["host_pointer_%s_pairs_per_s" % name] = r.get("pairs_per_s")
        if "ms_per_ipopt_iter" in out:
            shaped["ms_per_ipopt_iter"] = {k: v for k, v in out["ms_per_ipopt_iter"].items() if k != "what"}
        out["config"]["ipopt_shaped"] = shaped

    # ---- every rank: the other splits, so that at N > 1 the collective path is measured too ----------------------------
    if extras:
        secs = {}
        K2 = max(20, min(args.steps, 200))
        plan = [
            ("config3_intervals_b1", lambda: problems.launch(args.intervals, args.nodes), 1, 128, "perturb"),
            ("config3_intervals_b16", lambda: problems.launch(args.intervals, args.nodes), 16, 256, "perturb"),
            ("config4_hypersensitive_hp_intervals_b1", lambda: problems.config("hypersensitive"), 1, 256, "uniform"),
            ("config4_hypersensitive_hp_intervals_b64", lambda: problems.config("hypersensitive"), 64, 1024, "uniform"),
        ]
        for name, mk, b, r, mode in plan:
            try:
                secs[name] = device_workload(ctx, args, mk(), b, r, K2, 10, True, 3, mode=mode, use_graph=not args.no_graph)
            except Exception as ex:
                errors[name] = repr(ex)
        if out is not None:
            out["strong_scaling"] = {
                "what": "ONE problem's mesh intervals sharded over the %d rank(s); every rank ends each step with the complete g and "
                        "Jacobian of every iterate (device consumer); pairs_per_s is whole-job" % world,
                "workloads": secs}
        # config 5: the 1024-instance MPC sweep sharded by instance (fixed total work)
        try:
            from lpopc_amd.dist import shard_instances
            _, cnt = shard_instances(1024, rank, world)
            c5 = device_workload(ctx, args, problems.quadrotor(8, 8), cnt, 4 * cnt, K2, 10, False, 5 + 100000 * rank,
                                 use_graph=not args.no_graph)
            # whole-job rate: all ranks' instances over the max-over-ranks time
            c5["pairs_per_s_whole_job"] = 1024 * 1e3 / c5["ms_per_step"]
            c5["instances_total"] = 1024
            if out is not None:
                out["config5_mpc_sweep_instances_sharded"] = c5
        except Exception as ex:
            errors["config5"] = repr(ex)
        if world > 1:
            try:
                hc = host_consumer_section(ctx, args, prob, 1)
                if out is not None:
                    out["host_consumer"] = hc
            except Exception as ex:
                errors["host_consumer"] = repr(ex)

    # ---- ONE process driving every device (rpm_group_*): rank 0 alone, the other ranks idle on the host meanwhile -------------
    gdev = [int(t) for t in args.group_devices.split(",") if t.strip() != ""] or (list(range(world)) if world > 1 else [])
    if extras and gdev:
        store = None
        if ctx.dist is not None:
            ctx.barrier_sync()   # every rank's GPU is idle from here on
            try:
                from torch.distributed.distributed_c10d import _get_default_store
                store = _get_default_store()
            except Exception:
                store = None
        if rank == 0:
            try:
                out["single_process_group"] = group_section(ctx, args, prob, gdev)
            except Exception as ex:
                errors["single_process_group"] = repr(ex)
            if store is not None:
                store.set("rpm_group_section_done", "1")
        elif store is not None:
            store.wait(["rpm_group_section_done"])   # a host-side wait: no kernel spins on this rank's GPU
        if ctx.dist is not None:
            ctx.barrier_sync()

    if rank == 0:
        if extras and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(prob, xs_main, args.cpu_seconds)
            if out["cpu_baseline"] and "config" in out:
                cb = out["cpu_baseline"]
                out["cpu_baseline"]["gpu_over_cpu"] = {
                    "batched_device_resident": out["value"] / cb["value"],
                    "note": "reported ratio, not a quality measure (the roofline fraction is); per-call figures: config.ipopt_shaped"}
        else:
            out["cpu_baseline"] = None
        if errors:
            out["errors"] = errors
        print(json.dumps(out))
    if ctx.dist is not None:
        ctx.dist.barrier()
        ctx.dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py — eval_g+eval_jac_g throughput on the 4-phase, 4096-LGR-point Delta-III problem.

A "step" is one (eval_g, eval_jac_g) pair at a fresh NLP iterate x_k (new_x = true), evaluated by the
fused HIP pair kernel with x already resident in HBM.  R distinct seeded iterates are kept in HBM and
cycled.  At --gpus N > 1 every rank evaluates its own stream of iterates (independent problem
instances: multi-start / MPC-sweep sharding, no data-path collective), so the scaling is weak and
`value` is N*K pairs over the max-over-ranks time.  `--shard intervals` instead splits ONE problem's
mesh intervals over the ranks and all-gathers g / values with RCCL (strong scaling).

Contract line (one JSON object on stdout, rank 0):
  metric/unit  BASELINE.json's metric: eval_g+eval_jac_g pairs per second
  roofline     HBM roofline of the dominant kernel rpm_tile_kernel: algorithmic bytes per launch
               (8(2n+m+nnz)+8 sum N_k(N_k+1)) / average launch duration measured with HIP
               events on the launch stream over the timed region
  cpu_baseline the CPU oracle (a C port of lpopc's algorithm, oracle/) timed on one host core on a
               bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes(eng):
    """read x once per callback, write g, write every Jacobian value, read each D tile once."""
    d_tiles = 0
    for p in range(eng.n_phases):
        t = eng.phase_tables(p)
        d_tiles += t["d_vals"].size
    return 8 * (2 * eng.n + eng.m + eng.nnz_jac) + 8 * d_tiles


def cpu_baseline(prob, xs, budget_s):
    """The CPU oracle (C port of the reference algorithm, 1 thread) on a bounded sample."""
    from oracle.oracle import Oracle
    orc = Oracle(prob)
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
        if el >= budget_s and pairs >= 20:
            break
    return {"value": pairs / el, "unit": "pairs/s", "cores": 1, "kind": "port",
            "sample": "%d (eval_g,eval_jac_g) pairs of the same workload in %.1f s, oracle/liborpm.so -O2, 1 thread"
                      % (pairs, el)}


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
    ap.add_argument("--shard", choices=["instances", "intervals"], default="instances")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile", action="store_true",
                    help="profiling run: only the warm-up and the timed launches (no extra checks/sections), so that\n"
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
    ap.add_argument("--intervals", type=int, default=64)
    ap.add_argument("--nodes", type=int, default=16)
    args = ap.parse_args()

    import numpy as np
    import torch
    from lpopc_amd import problems
    from lpopc_amd.engine import NLPEngine

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    dist = None
    if world > 1 or "RANK" in os.environ:   # under torch.distributed.run: always go through the RCCL path
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)

    prob = problems.launch(args.intervals, args.nodes)
    sharded = args.shard == "intervals" and world > 1
    B = max(1, args.batch)
    if sharded and B != 1:
        raise SystemExit("--shard intervals evaluates one iterate per launch")
    eng = NLPEngine(prob, n_instances=B, shard_mode=1 if sharded else 0, shard_rank=rank if sharded else 0,
                    shard_world=world if sharded else 1, tile_nodes=args.tile_nodes, device=local_rank, role_loop=args.role_loop)
    if args.dx_mode:
        eng.set_option("dx_mode", args.dx_mode)
    if args.pipeline != -1:
        eng.set_option("pipeline", args.pipeline)
    xl, xu, _, _ = eng.get_bounds_info()
    x0 = eng.get_starting_point()
    R = max(args.iterates, 2 * B)
    R -= R % B
    # rank-specific iterates in the weak-scaling mode, identical ones when one problem is sharded
    seed0 = 3 if sharded else 3 + 1000 * rank
    xs = [problems.seeded_iterate(x0, xl, xu, seed0 + r) for r in range(R)]
    d_x = torch.from_numpy(np.stack(xs)).cuda()
    # every iterate's g / values array starts on a 128-byte boundary, as separately allocated arrays would: packed
    # back to back (m and nnz_jac are not multiples of 8) three quarters of the store runs would straddle 64-byte
    # granules, which costs the HBM write path a third of its rate (tools/ubench/store_pattern.py)
    if not sharded:
        eng.set_option("instance_align", args.instance_align)
    sg, sv = eng.get_option("stride_g"), eng.get_option("stride_values")
    d_g = torch.empty((R, sg), dtype=torch.float64, device="cuda")
    d_v = torch.empty((R, sv), dtype=torch.float64, device="cuda")
    comm = None
    if sharded:
        from lpopc_amd.dist import IntervalGather
        comm = IntervalGather(eng, dist, world)

    def step(k):
        r = (k * B) % R          # this step's batch of B consecutive resident iterates
        if args.unfused:
            eng.eval_g_dev(d_x[r], d_g[r])
            eng.eval_jac_g_dev(d_x[r], d_v[r])
        else:
            eng.eval_pair_dev(d_x[r], d_g[r], d_v[r])
        if comm is not None:
            comm.all_gather(d_g[r], d_v[r])

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    sync_all()

    use_graph = not args.no_graph and comm is None
    graph = None
    if use_graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step(0)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for k in range(args.steps):
                step(k)
        graph.replay()  # one untimed replay (graph upload)
        sync_all()

    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    sync_all()
    t0 = time.perf_counter()
    ev0.record()
    if graph is not None:
        graph.replay()
    else:
        for k in range(args.steps):
            step(k)
    ev1.record()
    sync_all()
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    if dist is not None:
        t = torch.tensor([elapsed, dev_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, dev_ms = float(t[0]), float(t[1])

    # sanity: results of the timed region are finite and equal to a fresh single evaluation
    if comm is None and not os.environ.get("RPM_DIAG_MASK") and not args.profile:
        one = NLPEngine(prob, tile_nodes=args.tile_nodes, device=local_rank)
        if args.dx_mode:
            one.set_option("dx_mode", args.dx_mode)
        chk_g = torch.empty(eng.m, dtype=torch.float64, device="cuda")
        chk_v = torch.empty(eng.nnz_jac, dtype=torch.float64, device="cuda")
        for r in (0, R - 1):
            one.eval_pair_dev(d_x[r], chk_g, chk_v)
            torch.cuda.synchronize()
            assert torch.equal(chk_g, d_g[r, :eng.m]) and torch.equal(chk_v, d_v[r, :eng.nnz_jac]) and bool(torch.isfinite(chk_v).all())
        one.close()

    if rank == 0:
        units_per_step = B * (1 if sharded else world)
        pairs = args.steps * units_per_step
        value = pairs / elapsed
        bytes_per_launch = algorithmic_bytes(eng) * B
        launches_per_step = 2 if args.unfused else 1
        launch_us = dev_ms * 1e3 / (args.steps * launches_per_step)
        achieved = bytes_per_launch / (dev_ms * 1e-3 / args.steps) / 1e9
        traffic = None
        # HBM bytes per launch from the rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs) of this
        # exact command, committed under profiles/ (tools/collect_profiles.sh regenerates it)
        tp = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tp) and B == 64 and not args.unfused and args.intervals == 64 and args.nodes == 16:   # the default command
            try:
                traffic = json.load(open(tp)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "eval_g+eval_jac_g calls/sec, 4-phase 4096-LGR-pt problem",
            "value": value,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if sharded else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "Delta-III 4-phase launch ascent, %d intervals/phase x %d LGR points (n=%d, m=%d, nnz_jac=%d), "
                            "first-derive=finite-difference tol=1e-6, %d seeded iterates resident in HBM, %d iterate(s) per step"
                            % (args.intervals, args.nodes, eng.n, eng.m, eng.nnz_jac, R, B),
                "pair": "unfused: eval_g kernel + eval_jac_g kernel" if args.unfused else
                        "fused: one tile-kernel launch writes g and all Jacobian values of the step's iterates",
                "launch": "hipGraph replay of the K steps" if graph is not None else "eager launches",
                "tile_nodes": eng.get_option("tile_nodes"),
                "thread_layout": ("persistent workgroups of two halves, each 4 compute waves + 2 DMA waves (rpm_tile_pl_kernel)" if eng.get_option("pipeline_active")
                                  else "64 nodes x 4 role groups (roles looped)" if eng.get_option("role_loop") else "16 nodes x (nx+nu+2) roles"),
                "dx_mode": "mfma_f64_16x16x4" if args.dx_mode else "scalar, reference summation order",
                "parallelism": ("intervals sharded x%d + RCCL all-gather" % world) if sharded else
                               ("independent instances x%d" % world),
                "ms_per_ipopt_iter_synthetic": None,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "kernel": ("rpm_tile_pl_kernel" if eng.get_option("pipeline_active") else
                           "rpm_tile_rl_kernel" if eng.get_option("role_loop") else "rpm_tile_kernel"),
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "algorithmic_bytes_per_pair": bytes_per_launch // B,
                "avg_launch_us": launch_us,
            },
        }
        # synthetic "ms per IPOPT iteration": one each of eval_f, eval_grad_f, eval_g, eval_jac_g at one x
        d_obj = torch.empty(B, dtype=torch.float64, device="cuda")
        d_grad = torch.empty((B, eng.n), dtype=torch.float64, device="cuda")
        if comm is None and not args.profile:
            torch.cuda.synchronize()
            ti = time.perf_counter()
            nit = 200
            for k in range(nit):
                r = (k * B) % R
                eng.eval_f_dev(d_x[r], d_obj)
                eng.eval_grad_f_dev(d_x[r], d_grad)
                eng.eval_pair_dev(d_x[r], d_g[r], d_v[r])
            torch.cuda.synchronize()
            out["config"]["ms_per_ipopt_iter_synthetic"] = (time.perf_counter() - ti) * 1e3 / (nit * B)
        if comm is None and B > 1 and not args.profile:
            # the same kernel with ONE iterate per launch (what a sequential Ipopt loop sees): latency-bound
            one = NLPEngine(prob, tile_nodes=args.tile_nodes, device=local_rank)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            g1 = torch.cuda.CUDAGraph()
            for k in range(20):
                one.eval_pair_dev(d_x[k % R], d_g[k % R], d_v[k % R])
            torch.cuda.synchronize()
            with torch.cuda.graph(g1):
                for k in range(256):
                    one.eval_pair_dev(d_x[k % R], d_g[k % R], d_v[k % R])
            g1.replay()
            torch.cuda.synchronize()
            e0.record()
            g1.replay()
            e1.record()
            torch.cuda.synchronize()
            us1 = e0.elapsed_time(e1) * 1e3 / 256
            out["config"]["single_iterate_per_launch"] = {
                "pairs_per_s": 1e6 / us1, "launch_us": us1,
                "hbm_frac": algorithmic_bytes(one) / (us1 * 1e-6) / 1e9 / HBM_PEAK_GBS}
            one.close()
        if not args.no_cpu_baseline and not args.profile:
            out["cpu_baseline"] = cpu_baseline(prob, xs, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Randomised cross-check of the three tile-kernel layouts (one-role, role-looped, pipelined): random ragged meshes,
random instance counts, bit-identical g and Jacobian required.  Run on the GPU box: python tools/stress_layouts.py [n]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from lpopc_amd import problems
from lpopc_amd.engine import NLPEngine

rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 30
makers = [("launch", problems.launch), ("bryson_denham", problems.bryson_denham), ("quadrotor", lambda: problems.quadrotor(2, 4)),
          ("climb", lambda: problems.min_time_climb(2, 4)), ("hypersensitive", lambda: problems.hypersensitive())]
bad = 0
for t in range(trials):
    name, mk = makers[rng.integers(len(makers))]
    prob = mk()
    for i in range(prob.GetPhaseNum()):
        K = int(rng.integers(1, 7))
        cuts = np.sort(rng.uniform(-0.95, 0.95, K - 1))
        ph = prob.GetPhase(i)
        ph.meshpoints = [-1.0] + [float(c) for c in cuts] + [1.0]
        ph.nodesperinterval = [int(v) for v in rng.integers(2, 20, K)]
    B = int(rng.integers(1, 400))
    base = NLPEngine(prob, device=0, role_loop=0)
    xl, xu, _, _ = base.get_bounds_info()
    x0 = base.get_starting_point()
    xs = np.stack([problems.seeded_iterate(x0, xl, xu, 1000 * t + b) for b in range(min(B, 3))])
    ref_g, ref_v = [], []
    for x in xs:   # eval_jac_g(new_x = False) returns the Jacobian cached by the eval_g of the SAME x
        ref_g.append(base.eval_g(x).copy())
        ref_v.append(base.eval_jac_g(x, False).copy())
    base.close()
    xb = np.tile(xs[0], (B, 1))
    xb[: xs.shape[0]] = xs
    dx = torch.from_numpy(xb).cuda()
    for mode in ("rl", "pl"):
        eng = NLPEngine(prob, n_instances=B, device=0, role_loop=1)
        eng.set_option("pipeline", 1 if mode == "pl" else 0)
        # outputs sit between guard bands: an out-of-bounds store of a kernel shows up as a changed guard word
        GUARD = 4096
        raw_g = torch.full((B * eng.m + 2 * GUARD,), 7.25, dtype=torch.float64, device="cuda")
        raw_v = torch.full((B * eng.nnz_jac + 2 * GUARD,), 7.25, dtype=torch.float64, device="cuda")
        dg = raw_g[GUARD:GUARD + B * eng.m].view(B, eng.m)
        dv = raw_v[GUARD:GUARD + B * eng.nnz_jac].view(B, eng.nnz_jac)
        dg.fill_(float("nan"))
        dv.fill_(float("nan"))
        eng.eval_pair_dev(dx, dg, dv)
        torch.cuda.synchronize()
        for raw in (raw_g, raw_v):
            if not (bool((raw[:GUARD] == 7.25).all()) and bool((raw[-GUARD:] == 7.25).all())):
                bad += 1
                print("GUARD BAND TOUCHED", t, name, mode, "B", B)
        active = eng.get_option("pipeline_active")
        g, v = dg.cpu().numpy(), dv.cpu().numpy()
        ok = all(np.array_equal(g[b], ref_g[b]) and np.array_equal(v[b], ref_v[b]) for b in range(xs.shape[0]))
        ok = ok and np.array_equal(g[-1], ref_g[0] if B > xs.shape[0] else ref_g[B - 1]) and not np.isnan(v).any()
        if not ok:
            bad += 1
            for b in range(xs.shape[0]):
                dgm, dvm = g[b] != ref_g[b], v[b] != ref_v[b]
                if dgm.any() or dvm.any():
                    print("   inst", b, "g diff", int(dgm.sum()), "first", np.nonzero(dgm)[0][:4], "max", np.abs(g[b] - ref_g[b]).max(),
                          "| v diff", int(dvm.sum()), "first", np.nonzero(dvm)[0][:4], "max", np.nanmax(np.abs(v[b] - ref_v[b])), "nan", int(np.isnan(v[b]).sum()))
                    break
            print("MISMATCH", t, name, mode, "B", B, [p.nodesperinterval for p in (prob.GetPhase(i) for i in range(prob.GetPhaseNum()))])
        eng.close()
    print("trial %d %-14s B=%3d n=%d pipelined=%d ok" % (t, name, B, len(x0), active), flush=True)
print("FAILED" if bad else "ALL LAYOUTS BIT-IDENTICAL", bad)
sys.exit(1 if bad else 0)

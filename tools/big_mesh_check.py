"""Largest-size sanity run (not in the test-suite): Delta-III on 4 x 4096 intervals x 16 points (n = 2.6 M, 54 M Jacobian
entries), one iterate: the pipelined kernel against the one-role kernel, bit for bit.  Run on the GPU box."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from lpopc_amd import problems
from lpopc_amd.engine import NLPEngine

K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
prob = problems.launch(K, 16)
t0 = time.time()
a = NLPEngine(prob, device=0, role_loop=0)
print("n=%d m=%d nnz_jac=%d, set-up %.1f s" % (a.n, a.m, a.nnz_jac, time.time() - t0), flush=True)
xl, xu, _, _ = a.get_bounds_info()
x = problems.seeded_iterate(a.get_starting_point(), xl, xu, 5)
dx = torch.from_numpy(x).cuda()
out = []
for eng in (a, NLPEngine(prob, device=0, role_loop=1)):
    dg = torch.full((eng.m,), np.nan, dtype=torch.float64, device="cuda")
    dv = torch.full((eng.nnz_jac,), np.nan, dtype=torch.float64, device="cuda")
    eng.eval_pair_dev(dx, dg, dv)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        eng.eval_pair_dev(dx, dg, dv)
    e1.record()
    torch.cuda.synchronize()
    print("  layout: tile_nodes=%d pipelined=%d  %.1f us per pair" % (eng.get_option("tile_nodes"), eng.get_option("pipeline_active"),
                                                                     e0.elapsed_time(e1) * 1e3 / 5), flush=True)
    out.append((dg.cpu().numpy(), dv.cpu().numpy()))
    eng.close()
ok = np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1]) and not np.isnan(out[1][1]).any()
print("bit-identical" if ok else "MISMATCH")
sys.exit(0 if ok else 1)

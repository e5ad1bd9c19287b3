"""Device time of the exact-Hessian path (rpm_eval_h_dev) on the metric problem, next to the CPU oracle.
Run on the GPU box: python tools/bench_hessian.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from lpopc_amd import problems
from lpopc_amd.engine import NLPEngine
from lpopc_amd.problem import Options
from oracle.oracle import Oracle

opts = Options()
opts.SetStringValue("hessian-approximation", "exact")
prob = problems.config("launch")
eng = NLPEngine(prob, opts, device=0)
xl, xu, _, _ = eng.get_bounds_info()
x = problems.seeded_iterate(eng.get_starting_point(), xl, xu, 3)
lam = np.random.RandomState(1).uniform(-1, 1, eng.m)
i, j = eng.eval_h_structure()
nnz = i.size
dx, dl = torch.from_numpy(x).cuda(), torch.from_numpy(lam).cuda()
dv = torch.empty(nnz, dtype=torch.float64, device="cuda")
for _ in range(5):
    eng.eval_h_dev(dx, 1.0, dl, dv)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
K = 50
for _ in range(K):
    eng.eval_h_dev(dx, 1.0, dl, dv)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / K
orc = Oracle(prob, opts)
orc.hess_structure()
t0 = time.perf_counter()
hv = orc.eval_h(x, 1.0, lam)
cpu_ms = (time.perf_counter() - t0) * 1e3
err = np.abs(dv.cpu().numpy() - hv)
print("nnz_h %d: device %.1f us per eval_h (%.1f GB/s of output), CPU oracle %.1f ms (x%.0f); entries bit-equal %.2f %%, max abs diff %.2e"
      % (nnz, us, nnz * 8 / us / 1e3, cpu_ms, cpu_ms * 1e3 / us, 100.0 * np.mean(err == 0), err.max()))

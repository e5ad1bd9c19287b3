"""Experiment: BASELINE config 5 (1024 quadrotor instances per launch) under the library RPM_HIP_LIB names: time per launch,
and a digest of g and the Jacobian values for fixed inputs (builds are compared by it).  Run on the GPU box."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch

from lpopc_amd import problems
from lpopc_amd.engine import NLPEngine
from bench_configs import gpu_rate

name = sys.argv[1] if len(sys.argv) > 1 else "quadrotor"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
prob = problems.quadrotor(8, 8) if name == "quadrotor" else problems.launch(64, 16)
e = NLPEngine(prob, device=0, n_instances=B)
xl, xu, _, _ = e.get_bounds_info()
x0 = e.get_starting_point()
xs = np.stack([problems.seeded_iterate(x0, xl, xu, 7 + i, "perturb") for i in range(4 * B)])
g, v = e.eval_pair(xs[:B].reshape(-1))
print("digest", hashlib.sha256(g.tobytes() + v.tobytes()).hexdigest()[:16], "pipeline", e.get_option("pipeline_active"), flush=True)
for rep in range(3):
    rate, us = gpu_rate(e, xs, B, steps=200)
    byts = B * 8.0 * (e.n + e.m + e.nnz_jac)
    print("%s x %d: %.1f us per launch, %.2f M pairs/s, %.3f of 8 TB/s" % (name, B, us, rate / 1e6, byts / us / 8e6), flush=True)
e.close()

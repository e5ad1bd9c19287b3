"""Receding-horizon demo of row f-2: B quadrotors fly to their own targets; every control period all B optimal-control
problems (8 x 8 LGR, horizon 2 s) are re-solved on the device from the measured states, warm-started from the previous
solutions (iterates stay in HBM: solve_dev in/out), and the first part of each plan is applied to a simple simulation
of the same dynamics.  Prints per-step solve time and iteration counts, cold vs warm.
python tools/mpc_closed_loop.py [instances] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from lpopc_amd import problems
from lpopc_amd.engine import BatchedIPM, NLPEngine
from lpopc_amd.problem import Options

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
K, NK = 8, 8
o = Options()
o.SetStringValue("hessian-approximation", "exact")
rng = np.random.RandomState(2)
targets = rng.uniform(-1.5, 1.5, size=(B, 3))
base = problems.quadrotor(K, NK)
eng = NLPEngine(base, o, n_instances=B, device=0)
for b in range(B):
    eng.set_instance_constants(b, problems.quadrotor(K, NK, pref=tuple(targets[b])).GetOpimalProblemFuns().consts)
ipm = BatchedIPM(eng, tol=1e-6)
one = NLPEngine(base, o, device=0)
xl, xu, _, _ = one.get_bounds_info()
x_guess = one.get_starting_point()
N1 = K * NK + 1
x0_idx = np.array([i * N1 for i in range(12)])
tau = np.concatenate([one.phase_tables(0)["points"], [1.0]])          # LGR points + the end point, in [-1, 1]
horizon = 2.0
dt = 0.1
state = np.zeros((B, 12))
state[:, :3] = rng.uniform(-0.3, 0.3, size=(B, 3))
XL, XU = np.tile(xl, (B, 1)), np.tile(xu, (B, 1))
d_x = torch.from_numpy(np.tile(x_guess, (B, 1))).cuda()
t_at = (tau + 1.0) * horizon / 2.0
for step in range(steps):
    XL[:, x0_idx] = XU[:, x0_idx] = state
    ipm.set_all_bounds(XL, XU)
    if step == 1:                                     # warm starts: small barrier, do not push the previous solution away
        ipm.set_option("mu_init", 1e-4)
        ipm.set_option("bound_push", 1e-6)
        ipm.set_option("bound_frac", 1e-6)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = ipm.solve_dev(d_x)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0)
    X = d_x.cpu().numpy()
    # "plant": follow the planned state trajectory for dt (the plan is dynamically consistent to the mesh accuracy)
    for j in range(12):
        traj = X[:, j * N1:(j + 1) * N1]
        state[:, j] = np.array([np.interp(dt, t_at, traj[b]) for b in range(B)])
    dist = np.linalg.norm(state[:, :3] - targets, axis=1)
    print("step %2d  %7.1f ms for %d solves  iterations %d..%d  converged %d/%d  mean distance to target %.3f" % (
        step, ms, B, r["iterations"].min(), r["iterations"].max(), int((r["status"] <= 1).sum()), B, dist.mean()), flush=True)
    # shift the plan: the next problem starts where this one is after dt; keep the rest as the guess
    for j in range(12):
        traj = X[:, j * N1:(j + 1) * N1]
        X[:, j * N1:(j + 1) * N1] = np.stack([np.interp(np.minimum(t_at + dt, horizon), t_at, traj[b]) for b in range(B)])
    d_x.copy_(torch.from_numpy(X))

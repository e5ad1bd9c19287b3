"""Limited-memory device solve at size: the metric problem and the 1024-instance sweep (experiment)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from lpopc_amd import problems
from lpopc_amd.engine import NLPEngine, BatchedIPM

for name, prob, B in (("launch_8x8", problems.launch(8, 8), 1), ("launch_metric", problems.launch(64, 16), 1), ("quadrotor_sweep", problems.quadrotor(8, 8), 256)):
    e = NLPEngine(prob, n_instances=B, device=0)
    s = BatchedIPM(e, max_iter=3000)
    x0 = np.tile(e.get_starting_point()[:e.n], (B, 1))
    if B > 1:
        x0 = x0 * (1 + 1e-3 * np.random.RandomState(0).uniform(-1, 1, x0.shape))
    t = time.time()
    r = s.solve(x0)
    dt = time.time() - t
    st = s.stats()
    print(name, "status", np.bincount(r["status"], minlength=6).tolist(), "iters max", int(r["iterations"].max()), "obj0", float(r["obj"][0]),
          "mass" if "launch" in name else "", -float(r["obj"][0]) * 301454.0 if "launch" in name else "", "err max", float(r["kkt_error"].max()),
          "%.2fs" % dt, st, "%.2f ms/iter" % (1e3 * dt / max(1, st["iterations"])), s.info()["half_bandwidth"], flush=True)
    s.close(); e.close()

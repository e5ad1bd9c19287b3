"""Row f-2 stress: random ragged meshes x problems x batch sizes, the device interior-point solver against its CPU
restatement (status, objective, iteration count).  python tools/stress_ipm.py [trials] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401

from lpopc_amd import problems
from lpopc_amd.engine import BatchedIPM, NLPEngine
from lpopc_amd.problem import Options
from oracle import ipm_oracle
from oracle.oracle import Oracle

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
o = Options()
o.SetStringValue("hessian-approximation", "exact")
bad = 0
for trial in range(trials):
    K = int(rng.randint(1, 7))
    cuts = np.sort(rng.uniform(-0.9, 0.9, size=K - 1))
    mesh = [-1.0] + cuts.tolist() + [1.0]
    nodes = rng.randint(2, 13, size=K).tolist()
    kind = ["hypersensitive", "brachistochrone", "quadrotor", "bryson_denham"][int(rng.randint(0, 4))]
    if kind == "hypersensitive":
        prob = problems.hypersensitive(mesh, nodes, tf=float(rng.choice([10.0, 40.0])))
    else:
        prob = getattr(problems, kind)(1, 4)
        ph = prob.GetPhase(0)
        del ph.meshpoints[:], ph.nodesperinterval[:]
        problems.set_mesh(ph, mesh, nodes)
    B = int(rng.randint(1, 6))
    eng = NLPEngine(prob, o, n_instances=B, device=0)
    orc = Oracle(prob, o)
    x0 = orc.starting_point()
    ipm = BatchedIPM(eng, tol=1e-7, max_iter=300)
    r = ipm.solve(np.tile(x0, (B, 1)))
    ref = ipm_oracle.solve(orc, x0, tol=1e-7, max_iter=300)
    info = ipm.info()
    same = (r["status"] == ref["status"]).all()
    if ref["status"] == 0:
        same = same and np.max(np.abs(r["obj"] - ref["obj"])) <= 1e-6 * max(1.0, abs(ref["obj"]))
    print("trial %2d %-16s K=%d nodes=%s B=%d kkt=%d band=%d border=%d  dev status %s it %s  orc status %s it %d  %s" % (
        trial, kind, K, nodes, B, info["kkt_order"], info["half_bandwidth"], info["border"], sorted(set(r["status"].tolist())),
        sorted(set(r["iterations"].tolist())), ref["status"], ref["iterations"], "ok" if same else "MISMATCH"), flush=True)
    bad += 0 if same else 1
    ipm.close()
    eng.close()
print("MISMATCHES", bad, "of", trials)

"""Delta-III from starts perturbed by 1e-10 (relative): how often does the device solver end at the optimum?
python tools/ipm_delta3_ensemble.py [seeds] [option=value ...]"""
import json
import os
import subprocess
import sys

here = os.path.dirname(os.path.abspath(__file__))
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
extra = sys.argv[2:]
ok = tot = 0
for K, Nk in ((4, 8), (8, 8), (16, 8), (64, 16)):
    for seed in range(1, seeds + 1):
        r = subprocess.run([sys.executable, os.path.join(here, "ipm_delta3.py"), str(K), str(Nk), "3000", "-1"] + extra, capture_output=True, text=True,
                           timeout=400, env=dict(os.environ, IPM_PERTURB_SEED=str(seed)))
        d = json.loads(r.stdout.splitlines()[0])
        good = d["status"] in (0, 1) and abs(d["final_mass_kg"] - 7529.71) < 0.05
        ok += good
        tot += 1
        print("4x%dx%d seed %d  status %d  iterations %4d  restorations %d  final mass %.4f kg  E_0 %.1e  %.2f s%s" % (
            K, Nk, seed, d["status"], d["iterations"], d["restorations"], d["final_mass_kg"], d["kkt_error"], d["solve_s"], "" if good else "   <-- FAILED"), flush=True)
print("%d of %d converged" % (ok, tot))

"""Delta-III on a ladder of meshes with both barrier strategies (one line each): python tools/ipm_delta3_table.py [option=value ...]"""
import json
import os
import subprocess
import sys

here = os.path.dirname(os.path.abspath(__file__))
extra = sys.argv[1:]           # further solver options, e.g. bound_relax_factor=1e-6
for K, Nk in ((2, 6), (4, 8), (8, 8), (16, 8), (32, 16), (64, 16)):
    for mu in (0, 1):
        r = subprocess.run([sys.executable, os.path.join(here, "ipm_delta3.py"), str(K), str(Nk), "3000", "-1", "mu_strategy=%d" % mu] + extra,
                           capture_output=True, text=True, timeout=400)
        try:
            d = json.loads(r.stdout.splitlines()[0])
            print("4x%dx%d  mu_strategy %s  status %d  iterations %4d  restorations %d  final mass %.4f kg  E_0 %.1e  %.2f s  %.2f ms/iteration" % (
                K, Nk, "adaptive" if mu else "monotone", d["status"], d["iterations"], d["restorations"], d["final_mass_kg"], d["kkt_error"], d["solve_s"],
                d["ms_per_iteration"]), flush=True)
        except Exception as ex:
            print("4x%dx%d mu %d failed: %s %s" % (K, Nk, mu, ex, r.stderr[-300:]), flush=True)

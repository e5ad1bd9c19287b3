import sys, os, json, subprocess, itertools
from concurrent.futures import ThreadPoolExecutor
HERE = os.path.dirname(os.path.abspath(__file__))
mesh = sys.argv[1:3]
configs = json.loads(sys.argv[3])
seeds = int(sys.argv[4])
def run(job):
    name, cfg, seed = job
    cfg = dict(cfg, verbose=0, perturb_seed=seed)
    env = dict(os.environ, OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-u", os.path.join(HERE, "d3_x.py"), "launch", *mesh, json.dumps(cfg)], capture_output=True, text=True, env=env, timeout=3000)
    return name, seed, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]
jobs = [(n, c, s) for n, c in configs.items() for s in range(seeds)]
with ThreadPoolExecutor(4) as ex:
    for name, seed, line in ex.map(run, jobs):
        print(name, seed, line, flush=True)

#!/bin/bash
# the metric problem from starts perturbed by 1e-10 (IPM_PERTURB_SEED), seeds $1..$2, solver options after them
# usage on the GPU box: bash tests/experiments/d3_seeds.sh 9 24 [option=value ...]
a=$1; b=$2; shift 2
for s in $(seq $a $b); do
  IPM_PERTURB_SEED=$s timeout -k 10 200 python tools/ipm_delta3.py 64 16 3000 -1 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('seed $s status %d iterations %4d factorisations %4d restorations %d mass %.4f E_0 %.1e %.2f s' % (d['status'], d['iterations'], d['factorizations'], d['restorations'], d['final_mass_kg'], d['kkt_error'], d['solve_s']))"
done

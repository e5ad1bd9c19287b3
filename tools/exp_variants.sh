#!/bin/bash
# Runs bench.py against each experiment build named on the command line (lpopc_amd/csrc/librpm_exp_<name>.so)
# and the default library; one JSON line each into gpurun_out/exp_<name>.json.  Perf exploration only.
mkdir -p gpurun_out
python bench.py --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/exp_base.json 2>gpurun_out/exp_base.err || exit 1
for v in "$@"; do
  RPM_HIP_LIB=$PWD/lpopc_amd/csrc/librpm_exp_$v.so python bench.py --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/exp_$v.json 2>gpurun_out/exp_$v.err || exit 1
done
python - "$@" <<'PY'
import json, sys
for v in ["base"] + sys.argv[1:]:
    d = json.loads(open("gpurun_out/exp_%s.json" % v).read().strip().splitlines()[-1])
    print(v, d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY

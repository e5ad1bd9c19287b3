#!/bin/bash
# timeline + bench of the diag build under each ablation mask given on the command line (perf exploration only)
mkdir -p gpurun_out
for m in "$@"; do
  echo "== RPM_DIAG_MASK=$m"
  RPM_DIAG_MASK=$m python tools/trace_timeline.py 16 || exit 1
  RPM_DIAG_MASK=$m RPM_HIP_LIB=$PWD/lpopc_amd/csrc/librpm_hip_diag.so python bench.py --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench us/launch', d['ms_per_step']*1e3)" || exit 1
done

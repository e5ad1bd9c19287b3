#!/bin/bash
# per-dispatch kernel trace of the 1024-instance quadrotor sweep (tools/bench_ipm.py 1024 0 1); run on the GPU box, then
# python tools/ipm_sweep_trace_summary.py here
set -e
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rm -rf gpurun_out/${TAG}_ipm_trace
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${TAG}_ipm_trace -- python3 tools/bench_ipm.py 1024 0 1 > gpurun_out/${TAG}_ipm_trace.log 2>&1
ls gpurun_out/${TAG}_ipm_trace/*/

#!/bin/bash
# per-dispatch kernel trace of the metric problem solved on the device (tools/ipm_delta3.py 64 16); run on the GPU box, then
# python tools/ipm_sweep_trace_summary.py <tag>_d3 here
set -e
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rm -rf gpurun_out/${TAG}_d3_ipm_trace
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${TAG}_d3_ipm_trace -- python3 tools/ipm_delta3.py 64 16 3000 -1 > gpurun_out/${TAG}_d3_ipm_trace.log 2>&1
ls gpurun_out/${TAG}_d3_ipm_trace/*/

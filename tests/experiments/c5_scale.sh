#!/bin/bash
# time per launch of the quadrotor batch against the batch size (fixed cost vs per-tile cost)
for B in 256 512 768 1024 1536 2048 4096; do
  timeout -k 10 200 python -u tests/experiments/c5_time.py quadrotor $B 2>&1 | grep -v amdgpu.ids | tail -1 || exit 1
done

#!/bin/bash
# Algorithm IC with and without the hot start (option ic_hot_start), Delta-III on several meshes and the quadrotor sweep
set -e
out=gpurun_out/ic_hot.txt
: > $out
for m in "4 8" "8 8" "16 8" "32 16" "64 16"; do
  for hot in 0 1; do
    echo "mesh $m hot $hot" >> $out
    python tools/ipm_delta3.py $m 3000 -1 ic_hot_start=$hot >> $out 2>&1
  done
done
for hot in 0 1; do
  echo "mesh 64 16 limited-memory? no: exact, hot $hot ic_hot_min=1e-14" >> $out
  python tools/ipm_delta3.py 64 16 3000 -1 ic_hot_start=$hot ic_hot_min=1e-14 >> $out 2>&1
done

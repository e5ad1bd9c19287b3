#!/bin/bash
# config 5 (or "launch 64") timing + digest under each experiment library named on the command line, and the default one
prob=${PROB:-quadrotor}; B=${B:-1024}
for v in base "$@"; do
  L=$PWD/lpopc_amd/csrc/librpm_hip.so; [ "$v" != base ] && L=$PWD/lpopc_amd/csrc/librpm_exp_$v.so
  echo "== $v"
  RPM_HIP_LIB=$L timeout -k 10 120 python -u tests/experiments/c5_time.py $prob $B 2>&1 | grep -v amdgpu.ids || exit 1
done

#!/bin/bash
# CPU-side sanitizer pass (the GPU pool runs no sanitizers): the oracle's C restatement under AddressSanitizer + UBSan through the
# CPU tests that drive it, and the page-lock registry's mock-runtime test (also part of the CPU suite).  Run in the container:
#   bash tools/sanitize_cpu.sh
set -e
cd "$(dirname "$0")/.."
make -C oracle -B liborpm.so CFLAGS="-O1 -g -ffp-contract=off -fPIC -std=c11 -fsanitize=address,undefined -fno-omit-frame-pointer" > /dev/null
trap 'make -C oracle -B liborpm.so > /dev/null' EXIT
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
  python -m pytest tests/test_oracle_invariants.py tests/test_static_parameters.py tests/test_known_answers.py tests/test_pin_registry_cpu.py -q -m "not gpu" -x
# the product library's HOST code (set-up, Jacobian / Hessian structure, sharding, mesh refinement, KKT plan, C ABI) under ASan: an
# experiment build (device code is left alone: ASan needs xnack+ there), loaded by every CPU test through RPM_HIP_LIB
if [ "$1" = "--host" ]; then
  make -C lpopc_amd/csrc librpm_exp_asan.so EXPFLAGS="-fsanitize=address -fno-omit-frame-pointer -g" > /dev/null 2>&1
  RT=$(find /opt/rocm/lib/llvm/lib/clang -name "libclang_rt.asan-x86_64.so" | head -1)
  ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 LD_PRELOAD=$RT RPM_HIP_LIB=$PWD/lpopc_amd/csrc/librpm_exp_asan.so \
    python -m pytest tests/ -q -m "not gpu" --deselect "tests/test_pin_registry_cpu.py::test_registry_bookkeeping_against_a_mock_runtime"
  rm -f lpopc_amd/csrc/librpm_exp_asan.so
fi

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

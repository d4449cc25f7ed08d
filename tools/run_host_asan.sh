#!/bin/bash
# CPU sanitizer run of the host entry points: host_enum.cpp built with
# -fsanitize=address,undefined, exercised by tests/test_host_logic.py (fixtures of the reference + oracle).
set -e
cd "$(dirname "$0")/.."
make -C temfpy_amd/csrc asan
LIBASAN=$(g++ -print-file-name=libasan.so)
LD_PRELOAD="$LIBASAN" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
  TMF_ASAN_LIB="$PWD/temfpy_amd/libtemfpy_host_asan.so" python -m pytest tests/test_host_logic.py -q -x \
  --deselect tests/test_host_logic.py::test_library_exports_every_symbol

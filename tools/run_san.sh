#!/usr/bin/env bash
# Host-side exercise of libcaphn (symbol table, size queries, argument validation -- no GPU call) against the HOST-sanitized build
# (make -C hypernet-image-captioning_amd/csrc san: AddressSanitizer + UBSan on the host code; GPU ASAN is not available on the pool).
# torch is not imported: a Python that preloads the ASAN runtime spends minutes inside torch's import.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
make -C "$ROOT/hypernet-image-captioning_amd/csrc" -j8 san > /dev/null
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
CAPHN_LIB_PATH="$ROOT/hypernet-image-captioning_amd/caphn/libcaphn_san.so" LD_PRELOAD="$RT" \
  ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
  timeout -k 5 300 python3 "$ROOT/tools/san_abi_check.py"

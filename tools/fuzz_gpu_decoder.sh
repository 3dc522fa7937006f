#!/bin/bash
# tools/fuzz_gpu_decoder.py on the AddressSanitizer build of the CPU model of the kernels:  tools/fuzz_gpu_decoder.sh [iterations] [seed]
set -eu
cd "$(dirname "$0")/.."
make -s -j8 -C tools/hipemu SAN=address
RT=/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so
LD_PRELOAD="$RT $(gcc -print-file-name=libstdc++.so)" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 HIPEMU_SLACK=0 \
  python tools/fuzz_gpu_decoder.py "${1:-300}" "${2:-1}"

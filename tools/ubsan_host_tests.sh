#!/bin/bash
# Runs the CPU test suite against an UndefinedBehaviorSanitizer build of the host library (shifts, signed overflow and
# misaligned accesses in the bitstream / header parsers and the host kernels).  libubsan is preloaded into the
# uninstrumented interpreter; the regular build is restored afterwards.  Run from the repository root.
set -eu
UBSAN=$(gcc -print-file-name=libubsan.so)
STDCXX=$(gcc -print-file-name=libstdc++.so)
restore() { rm -f dali_amd/build/host_*.o; make -s -C dali_amd/host; }
trap restore EXIT
rm -f dali_amd/build/host_*.o
make -s -C dali_amd/host CXXFLAGS="-O1 -g -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -fno-omit-frame-pointer -fsanitize=undefined -fno-sanitize=vptr -I../../include -pthread" \
     $(cd dali_amd/host && ls *.cpp | sed 's|\(.*\)\.cpp|../build/host_\1.o|')
g++ -shared -fPIC -pthread -fsanitize=undefined -o dali_amd/lib/libdali_amd_host.so dali_amd/build/host_*.o -Ldali_amd/lib \
    -ldali_amd_kernels -lz -Wl,-rpath,'$ORIGIN'
LD_PRELOAD="$UBSAN $STDCXX" UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 python -m pytest tests -q -m "not gpu" -p no:cacheprovider "$@"

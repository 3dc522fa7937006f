#!/bin/bash
# tools/fuzz_host_decoders.py (or, with a second argument "readers", tools/fuzz_host_readers.py) against an AddressSanitizer
# build of the host library (restored afterwards):  tools/asan_fuzz.sh [iterations] [decoders|readers]
set -eu
ASAN=$(gcc -print-file-name=libasan.so)
STDCXX=$(gcc -print-file-name=libstdc++.so)
restore() { rm -f dali_amd/build/host_*.o; make -s -C dali_amd/host; }
trap restore EXIT
rm -f dali_amd/build/host_*.o
make -s -C dali_amd/host CXXFLAGS="-O1 -g -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -fno-omit-frame-pointer -fsanitize=address -I../../include -pthread" \
     $(cd dali_amd/host && ls *.cpp | sed 's|\(.*\)\.cpp|../build/host_\1.o|')
g++ -shared -fPIC -pthread -fsanitize=address -o dali_amd/lib/libdali_amd_host.so dali_amd/build/host_*.o -Ldali_amd/lib \
    -ldali_amd_kernels -lz -Wl,-rpath,'$ORIGIN'
SCRIPT=tools/fuzz_host_decoders.py
if [ "${2:-decoders}" = "readers" ]; then SCRIPT=tools/fuzz_host_readers.py; fi
LD_PRELOAD="$ASAN $STDCXX" ASAN_OPTIONS=detect_leaks=0 python $SCRIPT "${1:-300}"

#!/bin/bash
# Builds build_variants/libdali_amd_kernels_<name>.so: the kernel library with ONE source recompiled under extra -D flags
# (locally: hipcc cross-compiles, no GPU time).   bash tools/build_variant.sh NAME source.hip "-DX=1 -DY=2"
set -e
NAME=$1; SRC=$2; FLAGS=$3
R=$(cd $(dirname $0)/.. && pwd)
mkdir -p $R/build_variants /tmp/variant_$NAME
cd $R/dali_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -I../../include -Wno-unused-function $FLAGS -c $SRC -o /tmp/variant_$NAME/obj.o
OBJS=$(ls ../build/*.o | grep -v "/host_" | grep -v "/$(basename $SRC .hip).o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build_variants/libdali_amd_kernels_$NAME.so $OBJS /tmp/variant_$NAME/obj.o
echo built $NAME
